cd /root/repo
python - <<'PY' 2>&1 | grep -v "class \|amdgpu" | tail -40
import sys, time; sys.path.insert(0,'.')
from lorikeet_amd import HipPairHMMEngine, synthetic
b=synthetic.ragged(); e=HipPairHMMEngine(0)
for _ in range(3): e.compute(b)
t=time.perf_counter(); e.compute(b); print("host path ms", (time.perf_counter()-t)*1e3)
e.set_switch("trace",1); e.compute(b)
PY
