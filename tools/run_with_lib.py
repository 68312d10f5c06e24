"""Development helper: the hot path of bench.py with another build of the engine library
(argv[1]), set before anything imports the engine -- for rocprofv3 runs, which must start
the Python program itself (no env or shell wrapper in front of it)."""
import os
import runpy
import sys

os.environ["OBNSL_ENGINE_LIB"] = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py", "--only-hot-path", "--steps", "120", "--warmup", "20", "--settle-ms", "0"] + sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
