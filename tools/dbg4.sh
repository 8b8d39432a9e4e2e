for m in zero tap11 tap00 tap22 full; do
python tools/dbg4.py $m 2 /tmp/a.npy; python tools/dbg4.py $m 4 /tmp/b.npy
python -c "
import numpy as np
a,b=np.load('/tmp/a.npy'),np.load('/tmp/b.npy')
d=np.abs(a-b)   # [board, C, 8, 8]
print('$m: scale', round(float(np.abs(a).max()),3), 'max diff per board', np.round(d.max(axis=(1,2,3)),4))
bad=d[0]>1e-3
print('   board0: bad channels', np.nonzero(bad.any(axis=(1,2)))[0][:40], 'count', int(bad.any(axis=(1,2)).sum()))
print('   board0: bad pixel map (count of bad channels per pixel)'); print(bad.sum(axis=0))"
done
python -c "
import numpy as np
a,b=np.load('/tmp/a.npy'),np.load('/tmp/b.npy')
np.set_printoptions(precision=3, suppress=True, linewidth=200)
for c in (0, 16, 20, 48):
    print('ch', c, 'ref line7', a[0,c,7], ' nb4 line7', b[0,c,7], ' ref line6', a[0,c,6])
"
