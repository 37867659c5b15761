import sys, traceback
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_developed_state as t
fails = 0
for r in range(int(sys.argv[1])):
    for mode in ("faithful", "clean"):
        for n, steps in ((40, 600), (48, 450)):
            try:
                t.test_one_step_from_a_developed_state(mode, n, steps)
            except AssertionError as e:
                fails += 1
                print("FAIL", r, mode, n, str(e).splitlines()[0][:120], flush=True)
print("done, fails:", fails, flush=True)
