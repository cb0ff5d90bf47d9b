"""PTZ_BATCHER_TRACE log (/tmp/iba_trace.txt) of a 64-rig run -> where the second repetition's time inside the call goes."""
import re, sys, statistics as st
L = open('/tmp/iba_trace.txt').read().splitlines()
rc = [l for l in L if l.startswith('batcher round clients')]
rr = [l for l in L if l.startswith('batcher round ran')]
n = 146 if len(rr) >= 292 else len(rr) // 2
rc, rr = rc[-n:], rr[-n:]
t, v, k = [], [], []
for l in rc:
    m = re.search(r'last round ([\d.]+) ms \(last view request after ([\d.]+), last registration request after ([\d.]+)', l)
    t.append(float(m.group(1))); v.append(float(m.group(2))); k.append(float(m.group(3)))
ran = [float(re.search(r'ran ([\d.]+) ms', l).group(1)) for l in rr]
print('rounds %d: rounds ran %.1f ms (median %.2f), clients between rounds %.1f ms (median %.2f; last view request median %.2f, last registration request median %.2f)'
      % (n, sum(ran), st.median(ran), sum(t), st.median(t), st.median(v), st.median(k)))
print(L[-1][:400])
