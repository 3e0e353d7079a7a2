"""How much of the kernel-by-kernel (N > 1 fallback) step is the HOST?  Times the enqueue of one eager LightGlue train step
(no synchronisation inside: pure Python + ctypes + torch dispatch time) against the device time of the same step and the
replayed hipGraph.  python tools/probe/eager_host_time.py [matcher|pipeline]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "matcher"
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    from glue_factory_amd import lib
    from glue_factory_amd.synthetic import to_device
    torch.cuda.set_device(0)
    lib.load()
    for graph in (False, True):
        if what == "matcher":
            model, cpu_data = bench.build_matcher(args, 0, "lightglue")
            data = to_device(cpu_data, "cuda")
            stepper = bench.make_stepper(args, model, 0, allow_graph=graph)
            step = lambda: stepper(data)["total"]
        else:
            step, _, stepper = bench.make_pipeline_step(args, 0, 0, graph=graph)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        host, dev = [], []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append(t1 - t0)
            dev.append(t2 - t0)
        # back-to-back (the bench's way): the host runs ahead of the device where it can
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        b2b = (time.perf_counter() - t0) / 10
        print(f"{what} graph={graph}: host enqueue {1e3 * sorted(host)[5]:.2f} ms, enqueue+drain {1e3 * sorted(dev)[5]:.2f} ms, "
              f"back-to-back {1e3 * b2b:.2f} ms/step")
        stepper.close()
        del stepper, step
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
