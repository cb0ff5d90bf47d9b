import re,sys
L=[l for l in open('/tmp/iba_trace.txt') if l.startswith('batcher round clients')]
L=L[len(L)//2:]
v=[];k=[];f=[];t=[]
for l in L:
    m=re.search(r'last round ([\d.]+) ms \(last view request after ([\d.]+), last registration request after ([\d.]+), first after ([\d.]+)',l)
    t.append(float(m.group(1)));v.append(float(m.group(2)));k.append(float(m.group(3)));f.append(float(m.group(4)))
import statistics as st
print('rounds',len(L),'think median %.2f; last view req median %.2f; last krt req median %.2f; first krt median %.2f'%(st.median(t),st.median(v),st.median(k),st.median(f)))
