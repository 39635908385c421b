mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_matchers_percall.py tests/test_gpu_parity.py -x -q -m gpu -k "matchers_percall or bow or triangulation or hamming" > gpurun_out/r03/t.log 2>&1; tail -4 gpurun_out/r03/t.log
timeout -k 10 120 python tools/matcher_bench.py 2>/dev/null > gpurun_out/r03/matchers_cur.json
python - <<'PY'
import json
t=open('gpurun_out/r03/matchers_cur.json').read(); d=json.loads(t[t.index('{'):])
for k in ("gpu","gpu_host_phases_us","gpu_batch","gpu_resident","gpu_resident_host_phases_us","gpu_resident_batch","cpu_oracle","verified"): print(k,d.get(k))
PY
timeout -k 10 120 python tools/diag_match_stamps.py 2>&1 | grep total | tee gpurun_out/r03/stamps_cur.txt
