# dispatch gaps between the launches of the training step (GPU box, repo root): rocprofv3 kernel trace -> per-boundary idle time
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 40 --warmup 10 --timed-steps 0 --cpu-rays 0 --dropin-steps 0 --highres-frames 0 --render-frames 0 > $OUT/line.json 2> $OUT/err.log
cd $ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gaps/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
# the last 30 iterations: find the select kernels
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "select_rays_and_pixels" in n]
idx = idx[-31:]
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for a, b in zip(idx[:-1], idx[1:]):
    seq = rows[a:b + 1]
    for x, y in zip(seq[:-1], seq[1:]):
        key = short(x["Kernel_Name"]) + " -> " + short(y["Kernel_Name"])
        gaps[key].append((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3)
        durs[short(x["Kernel_Name"])].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in gaps.items():
    print(f"gap {sum(v)/len(v):8.2f} us  x{len(v)//30}  {k}"); tot += sum(v) / 30
for k, v in durs.items():
    print(f"dur {sum(v)/len(v):8.2f} us  {k}")
print("sum of gaps per step (us):", tot, " step period (us):", (int(rows[idx[-1]]["Start_Timestamp"]) - int(rows[idx[0]]["Start_Timestamp"])) / 30e3)
PY
python tools/benchsum.py gpurun_out/gaps/line.json | head -3
