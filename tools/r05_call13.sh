#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
t0=$(date +%s)
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench4.json 2> gpurun_out/r05_bench4.err
echo "bench rc $? in $(( $(date +%s) - t0 )) s"; tail -n 2 gpurun_out/r05_bench4.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench4.json').readline())
print(d['value'], d['roofline']['frac'], d['roofline']['us_per_frame'])
for k,v in d['other_workloads'].items():
    pc=v.get('parity_check') or {}
    print('%-16s us/frame %7.3f frac %.4f  err %s tol %s ok %s over %s/%s' % (k, v['us_per_frame'], v['frac'], pc.get('max_block_rel_err'), pc.get('tolerance'), pc.get('ok'), pc.get('frames_over_tolerance'), pc.get('frames_checked')))
PY
