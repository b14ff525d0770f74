import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contrastiveseg_amd  # noqa: F401  (sets MIOPEN_USER_DB_PATH to the shipped records)
print("MIOPEN_USER_DB_PATH", os.environ.get("MIOPEN_USER_DB_PATH"), os.listdir(os.environ["MIOPEN_USER_DB_PATH"]))
sys.argv = ["conv_probe"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_probe.py"), run_name="__main__")
