# HBM traffic of the two fused kernels on the tree as it is: two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE) over tools/prof_run.py -> gpurun_out/${JPEGAMD_ROUND:-r02}_pmc_hbm_traffic.{txt,json} (copied to profiles/)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$C
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pm_$C -- python3 /root/repo/tools/prof_run.py 8192 > /root/repo/gpurun_out/pm_$C.log 2>&1 || { tail -5 /root/repo/gpurun_out/pm_$C.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, os, re, collections
RND = os.environ.get("JPEGAMD_ROUND", "r02")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pm_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(\w+_image_kernel)", row["Kernel_Name"])
            if m and row["Counter_Name"] == c:
                acc[m.group(1)][c].append(float(row["Counter_Value"]))
out, lines = {}, ["HBM traffic per launch, 8192x8192 QF=50 (tools/prof_run.py), rocprofv3 --pmc, one counter per pass;",
                  "FETCH_SIZE and WRITE_SIZE count KiB; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction)", ""]
alg = 6 * 8192 * 8192
for k, d in acc.items():
    f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    b = (2 * f + w) * 1024
    out[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": b, "launches": len(d["FETCH_SIZE"]),
              "ratio_to_algorithmic": b / alg}
    lines.append(f"{k}: FETCH_SIZE {f:.0f} KiB x 2 + WRITE_SIZE {w:.0f} KiB = {b/1e6:.1f} MB per launch = {b/alg:.3f} x algorithmic ({alg/1e6:.1f} MB), {len(d['FETCH_SIZE'])} launches")
json.dump({"8192x8192_qf50": out}, open(f"/root/repo/gpurun_out/{RND}_pmc_hbm_traffic.json", "w"), indent=1)
open(f"/root/repo/gpurun_out/{RND}_pmc_hbm_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
