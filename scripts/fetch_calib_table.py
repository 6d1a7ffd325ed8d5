"""Join the two rocprofv3 --pmc passes of scripts/ubench/fetch_calib (FETCH_SIZE, WRITE_SIZE) with the byte counts the program prints.
usage: python scripts/fetch_calib_table.py <dir with fetch/ write/ bytes.csv> > profiles/r6_fetch_calibration.md"""
import csv
import glob
import sys

d = sys.argv[1]
known = {r["kernel"]: (int(r["unique_read_bytes"]), int(r["unique_write_bytes"])) for r in csv.DictReader(open(f"{d}/bytes.csv"))}


def counter(sub, name):
    out = {}
    for f in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0]
                out[k] = float(r["Counter_Value"]) * 1024.0   # KiB
    return out


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
print("| pattern | unique bytes read | FETCH_SIZE reported | read factor (true / reported) | unique bytes written | WRITE_SIZE reported | write factor |")
print("|---|---|---|---|---|---|---|")
for k, (rb, wb) in known.items():
    f, w = fetch.get(k), write.get(k)
    rf = f"{rb / f:.3f}" if f and rb > (1 << 20) else "-"
    wf = f"{wb / w:.3f}" if w and wb > (1 << 20) else "-"
    print(f"| {k} | {rb / 1e6:.1f} MB | {(f or 0) / 1e6:.1f} MB | {rf} | {wb / 1e6:.1f} MB | {(w or 0) / 1e6:.1f} MB | {wf} |")
