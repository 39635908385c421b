# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/ubench/fetch_calib.hip); run on the GPU box: bash tools/calib_fetch.sh
set -e
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib > $OUT/$c.stdout 2> $OUT/$c.err
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "calib")
known = json.loads([l for l in open(os.path.join(out, "FETCH_SIZE.stdout")) if l.startswith("{")][-1])
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(out, c, "*", "*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][c + "_KiB"] = sum(v) / len(v)
B = known["bytes"]
doc = {"buffer_bytes": B, "note": "each kernel streams the 512 MiB buffer once; factor = counter bytes / known bytes (rocprofv3 reports KiB)", "kernels": {}}
for k, v in sorted(res.items()):
    need = B
    e = {"known_bytes": need}
    if k.startswith("k_rd"):
        e["FETCH_SIZE_bytes"] = int(v.get("FETCH_SIZE_KiB", 0) * 1024); e["fetch_factor"] = round(e["FETCH_SIZE_bytes"] / need, 4)
    else:
        e["WRITE_SIZE_bytes"] = int(v.get("WRITE_SIZE_KiB", 0) * 1024); e["write_factor"] = round(e["WRITE_SIZE_bytes"] / need, 4)
    doc["kernels"][k] = e
json.dump(doc, open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
