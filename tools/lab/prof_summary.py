import csv, sys, glob, re
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    ms = float(r['TotalDurationNs']) / steps / 1e6
    tot += ms
    if ms > 0.02:
        n = r['Name']
        n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
        n = re.sub(r'\(anonymous namespace\)::', '', n)
        print(n[:70].ljust(70), str(int(r['Calls']) // steps).rjust(4), f"{ms:7.3f} ms", f"{float(r['AverageNs'])/1e3:8.1f} us")
print('total', round(tot, 3))
