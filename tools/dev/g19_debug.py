import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_gpu_train as T
for prec in sys.argv[1:]:
    try:
        w = T._g19_check(prec, 10.0, 10.0, 1.0)
        print(prec, {k: f"{v:.1e}" for k, v in sorted(w.items(), key=lambda kv: -kv[1])})
    except Exception as e:
        print(prec, "FAILED", repr(e)[:2000])
