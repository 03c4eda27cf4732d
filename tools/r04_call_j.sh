#!/bin/bash
# Round-4 GPU call J: base_to_event_map written from the walk in phase 3 (no device pair lists in the fused host call):
# parity, A/B against the committed build (configs[1], with and without the fused phase), bench line.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
timeout -k 10 150 python -m pytest tests -m gpu -x -q -k "scaling or fused or 111_reads or recalibrated or process or shim or fuzz_alignment" > $O/t_fused.log 2>&1; echo "tests rc=$?" >> $O/steps.txt; tail -3 $O/t_fused.log
timeout -k 10 90 python tools/ab_quick.py r04head=build/libabea_r04head.so r04new=f5c_amd/libabea_hip.so r04head2=build/libabea_r04head.so --launches 9 --scaling-launches 5 > $O/ab_10k.log 2> $O/ab_10k.err; echo "ab rc=$?" >> $O/steps.txt; cat $O/ab_10k.log
timeout -k 10 200 python bench.py --steps 8 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/steps.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04j/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "host", j["host_to_host"]["host_ms_per_step"], "bound", j["bound"])
print("fused", j["fused_scaling"]["mevents_per_s"], j["fused_scaling"]["ms_per_step"])
print("kernel_only", j.get("kernel_only"))
PY
cat $O/steps.txt
