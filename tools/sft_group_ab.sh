cd $GRAFT_REPO_ROOT
for g in -1 0 -1 0; do
python - <<P
import sys, runpy, json, io, contextlib
sys.path.insert(0, ".")
from vila_amd import _lib
_lib.load().vila_gemm_force_group($g)
sys.argv = ["bench.py", "--mode", "sft", "--steps", "4", "--warmup", "2"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("group", $g, "sft ms/step", d["ms_per_step"])
P
done
