#!/bin/bash
# Round-4 GPU call G: scaling_single as the last phase of abea_align_kernel — parity first, then the bench line.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "scaling or fused or 111_reads or recalibrated or process or shim or fuzz_alignment or submit" > $O/t_fused.log 2>&1; echo "tests rc=$?" >> $O/steps.txt; tail -4 $O/t_fused.log
timeout -k 10 330 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/steps.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04g/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "host", j["host_to_host"]["host_ms_per_step"])
print("fused", json.dumps(j.get("fused_scaling"))[:600])
print("kernel_only", j.get("kernel_only"), "roofline frac", j["roofline"]["frac"])
PY
cat $O/steps.txt
