"""Cost of the sharded code path on ONE GPU: the C1 window solved plainly and with a one-rank native communicator attached
(accumulate -> ncclAllReduce of one rank -> decide -> solve).  Prints microseconds per Gauss-Newton iteration for both."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (torch's HIP runtime first)

torch.cuda.init()
from dsopp_amd import capi, synthetic as syn  # noqa: E402


def rate(g, n=40):
    g.snapshot()
    for _ in range(5):
        g.restore()
        g.optimize()
    t0 = time.perf_counter()
    its = 0
    for _ in range(n):
        g.restore()
        its += g.optimize()[1]
    return (time.perf_counter() - t0) / its * 1e6


def main():
    win = syn.make_window(num_frames=7, num_points=2000, width=640, height=480, seed=1)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    plain = rate(g)
    comm = capi.Comm(0, 1, 0, lambda raw: raw)
    g.set_comm(comm)
    sharded = rate(g)
    g.set_deterministic(True)
    sharded_det = rate(g)
    g.set_comm(None)
    det = rate(g)
    print(f"us per GN iteration (incl. restore per solve): plain {plain:.1f}, one-rank collective path {sharded:.1f}, "
          f"same + deterministic build {sharded_det:.1f}, deterministic without collective {det:.1f}")
    g.close()
    comm.close()


if __name__ == "__main__":
    main()
