#!/usr/bin/env python
"""Build profiles/pmc_traffic.json (HBM bytes per launch, by bench.py kernel name) from rocprofv3 --pmc passes.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <out>/fetch -- python bench.py <PROFILE ARGS>
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <out>/write -- python bench.py <PROFILE ARGS>
    python tools/make_pmc_traffic.py <out>/fetch <out>/write --gt-frames 8 --frames-per-leg 3 > profiles/pmc_traffic.json

with PROFILE ARGS = --steps 20 --warmup 5 --cpu-rays 0 --dropin-steps 0 --highres-frames 0 --render-frames 1 (separate passes, as the
microarchitecture guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  hbm_bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies
a 128-B fill as 64 B; calibrated on the streaming optimizer traffic, see profiles/*_pmc.md) + WRITE_SIZE x 1024.
The full-frame render launches `render_forward_kernel<9,false,false>` for the dataset images first (--gt-frames launches), then
--frames-per-leg launches per fwd_render leg in bench.py's order (init_field, traversal); they are told apart by dispatch order."""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import short  # noqa: E402

csv.field_size_limit(1 << 30)

BENCH_NAMES = {
    "render_forward_kernel<9, false, true>": "render_forward[spec,save]",
    "render_forward_kernel<1, true, true>": "render_forward[diffuse,save]",
    "render_forward_pair_kernel<9>": "render_forward[spec+diffuse,save]",
    "render_forward_pair_kernel": "render_forward[spec+diffuse,save]",
    "render_emit_direct_pair_kernel": "render_backward_emit_direct[spec+diffuse]",
    "render_emit_direct_pair_kernel<9>": "render_backward_emit_direct[spec+diffuse]",
    "render_emit_direct_kernel<9, false>": "render_backward_emit_direct[spec]",
    "render_emit_direct_kernel<1, true>": "render_backward_emit_direct[diffuse]",
    "render_backward_kernel<9, false, 0>": "render_backward[sh2]",
    "render_backward_kernel<1, true, 0>": "render_backward[diffuse]",
    "brick_gather_kernel<9, true>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, true, true>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, true, true, false>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, true, true, false, 4>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, true, true, false, 4, false>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, true, true, false, 4, true>": "brick_accumulate_adam_mirror[sh2]",
    "brick_gather_kernel<9, true, true, false, 8, false>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, true, false, false, 8, false>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, false, true, false, 8, false>": "brick_accumulate[sh2]",
    "brick_gather_kernel<9, false, false, false, 8, false>": "brick_accumulate[sh2]",
    "brick_gather_kernel<1, false, true, false, 8, false>": "brick_accumulate[base]",
    "brick_gather_kernel<1, false, false, false, 8, false>": "brick_accumulate[base]",
    "brick_gather_kernel<9, true, true, false, 8>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, true, false, false, 8>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, false, true, false, 8>": "brick_accumulate[sh2]",
    "brick_gather_kernel<9, false, false, false, 8>": "brick_accumulate[sh2]",
    "brick_gather_kernel<1, false, true, false, 8>": "brick_accumulate[base]",
    "brick_gather_kernel<1, false, false, false, 8>": "brick_accumulate[base]",
    "brick_gather_kernel<9, true, false, false>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, false, true, false>": "brick_accumulate[sh2]",
    "brick_gather_kernel<9, false, false, false>": "brick_accumulate[sh2]",
    "brick_gather_kernel<1, false, true, false>": "brick_accumulate[base]",
    "brick_gather_kernel<1, false, false, false>": "brick_accumulate[base]",
    "brick_gather_kernel<9, true, false>": "brick_accumulate_adam[sh2]",
    "brick_gather_kernel<9, false>": "brick_accumulate[sh2]",
    "brick_gather_kernel<9, false, false>": "brick_accumulate[sh2]",
    "brick_gather_kernel<1, false>": "brick_accumulate[base]",
    "brick_gather_kernel<1, false, false>": "brick_accumulate[base]",
    "adam_kernel": "adam_step",
    "bin_offsets_kernel": "bin_offsets",
    "loss_and_offsets_kernel": "l1_loss_grad+bin_offsets[both]",
}


def per_dispatch(root, counter):
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--gt-frames", type=int, default=8)
    ap.add_argument("--frames-per-leg", type=int, default=3)
    ap.add_argument("--source", default="")
    ap.add_argument("--bench-args", default="", help="the bench.py flags of the profiled command (recorded in the table)")
    ap.add_argument("--merge-into", default="", help="existing pmc_traffic.json: only ADD the kernels it does not have yet (profiles of another step variant)")
    args = ap.parse_args()
    by = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
    for counter, root in (("FETCH_SIZE", args.fetch_dir), ("WRITE_SIZE", args.write_dir)):
        for _, name, val in per_dispatch(root, counter):
            by[counter][name].append(val)
    import hashlib

    hip = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thr3ed_atom_amd", "csrc", "relu_field_kernels.hip")
    out = {"_kernel_source_sha256": hashlib.sha256(open(hip, "rb").read()).hexdigest(), "_bench_args": args.bench_args,
           "_source": args.source or f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py; hbm_bytes = FETCH_SIZE*1024*2 (gfx950) + WRITE_SIZE*1024, median per launch"}

    def median(v):
        v = sorted(v)
        return 0.5 * (v[(len(v) - 1) // 2] + v[len(v) // 2]) if v else 0.0

    def entry(fetch, write):  # median per launch: a few launches of the same kernel on other inputs (harness probes) do not skew it
        f = median(fetch) * 1024 * 2
        w = median(write) * 1024
        return {"hbm_bytes_per_launch": int(f + w), "fetch_bytes_corrected": int(f), "write_bytes": int(w), "launches": len(fetch)}

    for kname, bname in BENCH_NAMES.items():
        if by["FETCH_SIZE"].get(kname) and by["WRITE_SIZE"].get(kname):
            if len(by["FETCH_SIZE"][kname]) < 5:
                continue  # (a harness probe on other inputs -- e.g. the sample-count probe of the fwd_render leg --, not a kernel of the timed steps)
            out[bname] = entry(by["FETCH_SIZE"][kname], by["WRITE_SIZE"][kname])
    # (frames: the ray-packet kernel where the library dispatches it -- the bench's 800 x 800 frames at 128^3 --, else the per-ray kernel)
    tile_names = [k for k in by["FETCH_SIZE"] if k.startswith("render_frame_tile_kernel<9") or k == "render_frame_tile_kernel<true>"]
    frame = tile_names[0] if tile_names else "render_forward_kernel<9, false, false>"
    f, w = by["FETCH_SIZE"].get(frame, []), by["WRITE_SIZE"].get(frame, [])
    g, n = args.gt_frames, args.frames_per_leg
    for i, leg in enumerate(("init_field", "traversal")):
        fs, ws = f[g + i * n : g + (i + 1) * n], w[g + i * n : g + (i + 1) * n]
        if fs and ws:
            out[f"render_forward[sh2,frame]:{leg}"] = entry(fs, ws)
    if args.merge_into:
        base = json.load(open(args.merge_into))
        for k, v in out.items():
            if k not in base and not k.startswith("_") and ":" not in k:
                v["from"] = args.source
                base[k] = v
        out = base
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
