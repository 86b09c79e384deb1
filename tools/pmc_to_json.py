"""Fold the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_kernels.py under each) into
profiles/rNN_pmc_traffic.json: bytes per launch of the two roofline kernels.

usage: python tools/pmc_to_json.py <FETCH_counter_collection.csv> <WRITE_counter_collection.csv> <out.json>"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    """{kernel name: {grid size: [values...]}} of one counter; rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB."""
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            scale = 2048.0 if counter == "FETCH_SIZE" else 1024.0      # KiB -> bytes, x2 gfx950 read correction
            out.setdefault(r["Kernel_Name"], {}).setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]) * scale)
    return out


def main():
    import bench
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) -- "
                     "python3 tools/pmc_kernels.py, 1x MI355X",
           "unit": "bytes per launch (counter in KiB).  fetch = 2 x FETCH_SIZE: on gfx950 the counter reports half of the "
                   "bytes of a streaming read (MI355X_MICROARCH.md, HBM section); calibrated here for this kernel's two "
                   "access widths with tools/micro/fetch_calib.hip (1 GiB streamed by 4 B/lane and by 16 B/lane loads: "
                   "0.50000 GiB reported both times).  write = WRITE_SIZE as reported (exact for streaming stores)",
           "fetch_size_correction": 2.0,
           "kernels": {}}
    # 3x3 stride-1 convs: one grid size per shape, launch mix of one perception pass
    shapes = {}
    for c in bench.resnet_conv_table(*bench.IMG):
        if c[3] == 1:
            shapes[c] = shapes.get(c, 0) + 1
    def lookup(table, grid):
        for kname, grids_ in table.items():
            if "conv2d_hs3x3" in kname and grid in grids_:       # conv2d_hs3x3_kernel<...> and conv2d_hs3x3q_kernel<...>
                return grids_[grid]
        raise KeyError(grid)

    grids = {}
    for (cin, cout, k, s, p, h, w), cnt in shapes.items():
        cols = (bench.B * (w + 1) - 1 + 31) // 32          # column tiles over the images side by side, one shared zero column between neighbours
        if cout % 128 == 0 and cin % 64 == 0:              # conv2d_hs3x3q_eligible: the 16x16x32 kernel, 8 rows x 32 columns x 128 channels, 512 threads
            wgs, nt, label = ((h + 7) // 8) * cols * (cout // 128), 512, "16x16x32 kernel"
        else:
            mode = 0 if cin < 256 else (1 if h > 8 else 2)       # conv2d_hs_launch's tile-mode rule
            th, ct, nt = (16 if mode == 1 else 8), (2 if mode == 2 else 1), (256 if mode == 0 else 512)
            wgs, label = ((h + th - 1) // th) * cols * (cout // (64 * ct)), f"tile mode {mode}"
        grids[wgs * nt] = (f"{cin}->{cout} @{h}x{w} ({label})", cnt)
    per_shape, tf, tw, n = {}, 0.0, 0.0, 0
    for g, (label, cnt) in grids.items():
        fv, wv = lookup(fetch, g), lookup(write, g)
        f_, w_ = sum(fv) / len(fv), sum(wv) / len(wv)
        per_shape[label] = {"count": cnt, "fetch": f_, "write": w_}
        tf += f_ * cnt
        tw += w_ * cnt
        n += cnt
    res["kernels"]["conv2d_hs3x3_kernel<0|1|2> + conv2d_hs3x3q_kernel"] = {"per_shape": per_shape, "fetch": tf / n, "write": tw / n,
                                                         "traffic": (tf + tw) / n}
    tname = next(k for k in fetch if "tconv_hs_kernel<2" in k)
    fv = [v for g in fetch[tname].values() for v in g]
    wv = [v for g in write[tname].values() for v in g]
    res["kernels"]["tconv_hs_kernel<2,8,4> 512->512 k5, 128x4 positions"] = {
        "fetch": sum(fv) / len(fv), "write": sum(wv) / len(wv), "traffic": sum(fv) / len(fv) + sum(wv) / len(wv)}
    with open(sys.argv[3], "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
