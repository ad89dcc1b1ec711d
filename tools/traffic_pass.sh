# HBM-side traffic of the bench's kernels from the PMC counters (MI355X_MICROARCH.md, "HBM" + "rocprofv3 PMC slots"):
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (3 + 2 TCC slots do not fit one pass), kernel trace only.
# Corrections applied by the summary: both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of
# wide (16 B/lane) coalesced reads at 64 B -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
# usage (on the GPU box): bash tools/traffic_pass.sh <tag>      -> gpurun_out/traffic_<tag>/{summary.csv,hbm_traffic.json}
TAG=${1:-r1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-engine --no-host-fed --no-probe --no-sustained --no-index-leg > $OUT/$C.log 2>&1 || exit 1
done
cd $R
GIT=${GIT_SHA:-unknown}
python3 - "$OUT" "$GIT" <<'PY'
import collections, csv, glob, hashlib, json, os, sys
out, git = sys.argv[1], sys.argv[2]
hsh = hashlib.sha256()          # the kernel sources this pass ran on: bench.py emits `traffic` only for a matching tree (bench.csrc_sha16)
for f in sorted(os.listdir("segmminterest_amd/csrc")):
    if f.endswith((".h", ".hip", ".inc")):
        hsh.update(f.encode() + b"\0" + open(os.path.join("segmminterest_amd/csrc", f), "rb").read())
csrc = hsh.hexdigest()[:16]
acc = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in sorted(glob.glob("%s/%s/**/*counter_collection.csv" % (out, c), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc.setdefault(name, {"FETCH_SIZE": [], "WRITE_SIZE": []})[c].append(float(r["Counter_Value"]))
rows = []
for name, v in acc.items():
    nf, nw = len(v["FETCH_SIZE"]), len(v["WRITE_SIZE"])
    rd = 2.0 * 1024.0 * sum(v["FETCH_SIZE"]) / max(nf, 1)      # KiB -> B, x2: gfx950 correction for wide reads
    wr = 1024.0 * sum(v["WRITE_SIZE"]) / max(nw, 1)
    rows.append((name, nf, nw, rd, wr))
rows.sort(key=lambda r: -(r[3] + r[4]) * r[1])
with open(out + "/summary.csv", "w") as o:
    o.write("kernel,dispatches_fetch_pass,dispatches_write_pass,read_bytes_per_dispatch(2x FETCH_SIZE KiB),write_bytes_per_dispatch(WRITE_SIZE KiB)\n")
    for r in rows:
        o.write("%s,%d,%d,%.0f,%.0f\n" % (r[0].replace(",", ";"), r[1], r[2], r[3], r[4]))
g = [r for r in rows if "gemm_split_mfma" in r[0] or "gemm_f32_mfma" in r[0] or "gemm_pl_" in r[0]]
engine = os.environ.get("SEGMM_GEMM", "f16x3p")
n = sum(r[1] for r in g)
tot = sum((r[3] + r[4]) * r[1] for r in g)
json.dump({"kernel": "every GEMM dispatch of the run (gemm_pl_nt / gemm_pl_tn / gemm_split_mfma / gemm_f32_mfma)", "engine": engine, "git": git, "csrc_sha16": csrc, "launches": n,
           "hbm_bytes_per_launch": tot / max(n, 1), "read_bytes_per_launch": sum(r[3] * r[1] for r in g) / max(n, 1),
           "write_bytes_per_launch": sum(r[4] * r[1] for r in g) / max(n, 1),
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 3 --warmup 1; "
                     "KiB -> bytes, FETCH_SIZE doubled (gfx950 wide-read correction); mean over every GEMM dispatch of the run"},
          open(out + "/hbm_traffic.json", "w"), indent=1)
print(open(out + "/summary.csv").read())
print(open(out + "/hbm_traffic.json").read())
PY
