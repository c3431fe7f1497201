# Two processes on ONE GPU at the same time: 2500 forward iterations at 8 chains (bit-identical results) beside 150 training
# steps at 32 chains -- the dataflow Cholesky launches of both interleave on the chip (run on the GPU box).
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/soak_a.txt 2>&1 &
import sys, os, time
sys.path.insert(0, os.getcwd())
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c2", S=8)
e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram")
e.set_data(Y, c); e.set_params(params)
first = e.nll_terms()
for i in range(2500):
    assert e.nll_terms()["nll"] == first["nll"], i
print("A ok", first["nll"])
PY
PA=$!
python - <<'PY' > gpurun_out/soak_b.txt 2>&1 &
import sys, os, time
sys.path.insert(0, os.getcwd())
from ffvd_amd import synthetic
from ffvd_amd.engine import ElboEngine
params, Y, c, meta = synthetic.make_named("c2", S=32)
e = ElboEngine(meta["T"], meta["D"], meta["C"], meta["M"], meta["S"], route="gram", grad=True)
e.set_data(Y, c); e.set_params(params)
nl = []
for i in range(150):
    nl.append(e.adam_step(1e-3)["nll"])
print("B ok", nl[0], nl[-1])
PY
PB=$!
wait $PA; RA=$?; wait $PB; RB=$?
echo rc $RA $RB; tail -n 2 gpurun_out/soak_a.txt; tail -n 2 gpurun_out/soak_b.txt; [ $RA = 0 ] && [ $RB = 0 ]
