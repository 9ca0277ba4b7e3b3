import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'nh_cluster_kernel_fast' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [ (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
n = len(d)//8*8
by = collections.defaultdict(list)
for i in range(n): by[i%8].append(d[i])
for k in range(8): 
    v = sorted(by[k]); print(k, "median %.2f us  min %.2f  n %d" % (v[len(v)//2], v[0], len(v)))
