"""Prints a steady-state window of a rocprofv3 --kernel-trace --memory-copy-trace run (csv): start / end / duration in us, stream, name."""
import csv, glob, re, sys
d = sys.argv[1]; span = float(sys.argv[2]) if len(sys.argv) > 2 else 1400.0; back = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ker = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
cop = list(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])))
ev = [(int(c['Start_Timestamp']), int(c['End_Timestamp']), 'COPY ' + c['Direction'][12:], c['Stream_Id']) for c in cop]
for k in ker:
    m = re.search(r'(k_[a-z0-9_]+|__amd_rocclr_\w+)', k['Kernel_Name'])
    if m: ev.append((int(k['Start_Timestamp']), int(k['End_Timestamp']), m.group(1) + ' g' + k['Grid_Size_X'], k['Stream_Id']))
ev.sort()
rem = [e for e in ev if 'k_remap' in e[2]]
t0 = rem[-back][0]
for s, e, n, st in ev:
    if t0 - 50_000 <= s < t0 + span * 1000:
        print("%8.1f %8.1f %6.1f  s%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, st, n))
