"""Per hardware queue: what the head phase of a timed forward looks like in a rocprofv3 kernel trace (bench.py --steps 3 --warmup 1 under
`rocprofv3 --kernel-trace`): first start, last end and busy time of every queue after the backbone's last kernel.
usage: python tools/head_phase_queues.py <kernel_trace.csv> [forward index]"""
import csv, collections, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'copyBuffer' not in r['Kernel_Name'] and 'fillBuffer' not in r['Kernel_Name']]
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if 'im2col_rows' in r['Kernel_Name']]
fi = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fw = rows[starts[fi]:starts[fi + 1]]
t0 = fw[0]['s']
last_bb = max(i for i, r in enumerate(fw) if 'attn_v4_kernel' in r['Kernel_Name'] or 'attn_v3_kernel' in r['Kernel_Name'])
hs = fw[last_bb + 1]['s']
# the backbone ends a few kernels after the last cross-view attention launch (combine, proj, LN, fc1, fc2): the head phase starts at the first
# kernel on another queue
mainq = fw[0]['Queue_Id']
hs = min(r['s'] for r in fw if r['Queue_Id'] != mainq)
print(f"forward {fi}: span {(max(r['e'] for r in fw) - t0) / 1e6:.2f} ms, head phase from {(hs - t0) / 1e6:.2f} ms = {(max(r['e'] for r in fw) - hs) / 1e6:.2f} ms")
for q in sorted(set(r['Queue_Id'] for r in fw)):
    qr = [r for r in fw if r['Queue_Id'] == q and r['s'] >= hs]
    if not qr: continue
    big = collections.Counter()
    for r in qr: big[re.sub(r'\(.*', '', re.sub(r'.*::', '', r['Kernel_Name']))[:28]] += (r['e'] - r['s']) / 1e6
    print(f"  queue {q}: {len(qr):3d} kernels, {(qr[0]['s'] - t0) / 1e6:6.2f} -> {(max(r['e'] for r in qr) - t0) / 1e6:6.2f} ms, busy {sum(r['e'] - r['s'] for r in qr) / 1e6:5.2f} ms;", ', '.join(f"{k} {v:.2f}" for k, v in big.most_common(3)))
