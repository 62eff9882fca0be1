"""Join tools/hbm_kernels.py's algorithmic bytes with rocprofv3's kernel stats and FETCH_SIZE / WRITE_SIZE passes.
usage: hbm_table.py <log with the JSON line> <kernel_stats.csv> <FETCH counter csv> <WRITE counter csv> <out.csv>
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide streaming reads); both counters are reported in KiB."""
import collections, csv, json, re, sys
log, stats_p, f_p, w_p, out_p = sys.argv[1:6]
alg = json.loads(re.findall(r'\{.*\}', open(log).read())[-1])
stats = {r['Name']: r for r in csv.DictReader(open(stats_p))}
def pmc(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}
F, W = pmc(f_p), pmc(w_p)
def pick(table, kname):
    base = kname.split('<')[0]
    ks = [n for n in table if base in n and ((('<true>' in n) == ('<true>' in kname)) if 'token_move' in kname else True)]
    return ks[0] if ks else None
lines = ["kernel,calls,avg_us,algorithmic_bytes,achieved_GBps,frac_of_8TBps,FETCH_SIZE_x2_bytes,WRITE_SIZE_bytes"]
for key, nbytes in alg.items():
    kname = key.split(' ')[0]
    n = pick(stats, kname); avg = float(stats[n]['AverageNs'])
    fk = pick(F, kname)
    lines.append(f"\"{key}\",{stats[n]['Calls']},{avg / 1e3:.1f},{nbytes},{nbytes / avg:.0f},{nbytes / avg / 8000:.3f},"
                 f"{F[fk] * 2048 if fk else 0:.0f},{W[fk] * 1024 if fk else 0:.0f}")
open(out_p, 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
