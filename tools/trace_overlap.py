"""Timeline of the last few steps from a rocprofv3 --kernel-trace CSV: start/end (us) per kernel,
to check that k3_nav overlaps k2_lidar and to read the inter-kernel gaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-22:])
      for r in rows if any(t in r["Kernel_Name"] for t in ("k1_", "k2_", "k23_", "k3_", "k_step", "copyBuffer"))]
ks.sort()
k1 = [i for i, k in enumerate(ks) if "k1_" in k[2]]
i0 = k1[-4]
t0 = ks[i0][0]
for s, e, n in ks[i0:i0 + 22]:
    print("%-24s start %8.1f  end %8.1f  dur %6.1f" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
per = [(ks[k1[i + 1]][0] - ks[k1[i]][0]) / 1e3 for i in range(len(k1) - 60, len(k1) - 1)]
print("step period (us): mean %.1f min %.1f max %.1f" % (sum(per) / len(per), min(per), max(per)))
