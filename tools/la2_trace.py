"""Timeline of the LAST factorisation of a rocprofv3 --kernel-trace of tools/la2_time.py (two-stream look-ahead):
per outer block the bulk launch (start, duration) and what the chain did meanwhile.  usage: la2_trace.py <dir>"""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last factorisation: from the last k_zero_words before the last k_mirror_lower
mir = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_mirror_lower')][-1]
start = max(i for i, r in enumerate(rows[:mir]) if r['Kernel_Name'].startswith('k_zero_words'))
sel = rows[start:mir + 1]
t0 = int(sel[0]['Start_Timestamp'])
def us(x): return (int(x) - t0) / 1e3
print('factorisation: %.1f us, %d launches' % (us(sel[-1]['End_Timestamp']), len(sel)))
short = lambda n: n.split('(')[0].replace('void ', '')[:28]
trail = [r for r in sel if 'k_ldlt_trailing_64' in r['Kernel_Name']]
for r in trail:
    a, b = us(r['Start_Timestamp']), us(r['End_Timestamp'])
    inside = [q for q in sel if q is not r and us(q['Start_Timestamp']) < b and us(q['End_Timestamp']) > a]
    kinds = {}
    for q in inside:
        k = short(q['Kernel_Name'])
        d = (int(q['End_Timestamp']) - int(q['Start_Timestamp'])) / 1e3
        kinds.setdefault(k, []).append(d)
    desc = ', '.join('%s x%d avg %.1f' % (k, len(v), sum(v) / len(v)) for k, v in kinds.items())
    print('trailing [%7.1f .. %7.1f] %6.1f us (queue %s) | overlapping: %s' % (a, b, b - a, r.get('Queue_Id', '?'), desc or '-'))
