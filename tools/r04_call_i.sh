#!/bin/bash
# Round-4 GPU call I (closing): whole GPU suite and the bench line of the build that ships, then the fused call's kernel timeline
# (scaling_single inside the alignment kernel) for comparison with profiles/r04/e_fused_20k_kernel_trace.csv.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/steps.txt; tail -14 $O/gpu_tests.log
timeout -k 10 330 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/steps.txt
timeout -k 10 150 python tools/fused_trace.py 20000 /tmp/ft > $O/ft_gen.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused -o fused -- python tools/fused_trace.py 20000 /tmp/ft > $O/fused.log 2>&1
echo "fused trace rc=$?" >> $O/steps.txt; grep -v "^[WE]2026" $O/fused.log
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04i/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "host", j["host_to_host"]["host_ms_per_step"], "bound", j["bound"])
print("fused", j["fused_scaling"]["mevents_per_s"], j["fused_scaling"]["ms_per_step"])
print("kernel_only", j.get("kernel_only"), "roofline", j["roofline"]["frac"], j["roofline"]["valu_roofline"]["frac"])
PY
cat $O/steps.txt
