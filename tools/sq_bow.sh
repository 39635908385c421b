cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/sqb; timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/sqb -- python3 $GRAFT_REPO_ROOT/bench.py --workload euroc_bow --steps 3 --warmup 1 --cpu-frames 0 --extras 0 --no-verify > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/sqb/*/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0].replace("void ","")
    if n.startswith("k_bow"): acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,cs in acc.items(): print(n, {c: round(sum(v)/len(v)) for c,v in cs.items()})
PY
done
