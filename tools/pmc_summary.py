"""Summarise rocprofv3 --pmc passes over bench.py into the tables kept under profiles/.
usage: pmc_summary.py <util counter csv> <FETCH counter csv> <WRITE counter csv> <out prefix (e.g. profiles/r02)>"""
import collections, csv, json, sys
util_p, f_p, w_p, prefix = sys.argv[1:5]
CMD = "bench.py --steps 2 --warmup 1 --no-cpu-baseline"


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[r["Kernel_Name"]][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return per, dur


per, dur = load(util_p)
rows = []
for k, c in per.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) == 0:
        continue
    n = len(c["GRBM_GUI_ACTIVE"])
    avg_ns = sum(dur[k].values()) / len(dur[k])
    gui = sum(c["GRBM_GUI_ACTIVE"]) / n
    clock = gui / 8 / avg_ns
    busy = (sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / n) / (gui / 8 * 1024)
    rows.append((sum(dur[k].values()), k, n, avg_ns / 1e6, clock, busy, busy * clock / 2.4))
rows.sort(reverse=True)
with open(prefix + "_pmc_mfma_util.csv", "w") as fh:
    fh.write(f"# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over: {CMD}\n"
             "# clock_ghz = GRBM_GUI_ACTIVE / 8 (XCDs) / duration; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs);\n"
             "# frac_of_2p4ghz_peak = mfma_busy_frac * clock_ghz / 2.4 (comparable with bench.py's roofline.frac)\n"
             "kernel,launches,avg_ms,clock_ghz,mfma_busy_frac,frac_of_2p4ghz_peak\n")
    for _, k, n, ms, clock, busy, frac in rows:
        fh.write(f"\"{k[:100]}\",{n},{ms:.4f},{clock:.3f},{busy:.4f},{frac:.4f}\n")
        print(f"{k[28:90]:62s} n={n:4d} {ms:.4f} ms clock {clock:.2f} busy {busy:.3f} frac {frac:.3f}")

F, _ = load(f_p)
W, _ = load(w_p)
with open(prefix + "_pmc_hbm_traffic.csv", "w") as fh:
    fh.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over: {CMD}\n"
             "# bytes per launch = average over launches; FETCH_SIZE (KiB) x 1024 x 2 (gfx950 counts 128-byte requests as 64 bytes), WRITE_SIZE (KiB) x 1024\n"
             "kernel,launches,fetch_bytes_per_launch,write_bytes_per_launch\n")
    out = []
    for k in F:
        if k not in W or "anonymous namespace" not in k:
            continue
        f = sum(F[k]["FETCH_SIZE"]) / len(F[k]["FETCH_SIZE"]) * 2048
        w = sum(W[k]["WRITE_SIZE"]) / len(W[k]["WRITE_SIZE"]) * 1024
        out.append((k, len(F[k]["FETCH_SIZE"]), f, w))
    for k, n, f, w in sorted(out, key=lambda r: -r[1] * (r[2] + r[3])):
        fh.write(f"\"{k[:100]}\",{n},{f:.0f},{w:.0f}\n")
dom = [r for r in out if "gemm_nt_quad_kernel<0, false, 0>" in r[0]][0]
json.dump({"kernel": "gemm_nt_quad_kernel<0, false, 0>", "traffic_bytes_per_launch": int(dom[2] + dom[3]),
           "fetch_bytes_per_launch": int(dom[2]), "write_bytes_per_launch": int(dom[3]),
           "source": prefix + "_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, gfx950 correction applied)"},
          open("profiles/hbm_traffic.json", "w"), indent=1)
print(open("profiles/hbm_traffic.json").read())
