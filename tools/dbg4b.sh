python tools/dbg4.py zero 2 /tmp/a.npy; python tools/dbg4.py zero 4 /tmp/b.npy
python -c "
import numpy as np
a,b=np.load('/tmp/a.npy'),np.load('/tmp/b.npy')
np.set_printoptions(precision=3, suppress=True, linewidth=220)
print('nonfinite count', (~np.isfinite(b)).sum(), 'of', b.size)
nf=~np.isfinite(b[0])
print('board0 nonfinite channels', np.nonzero(nf.any(axis=(1,2)))[0][:64])
print('board0 nonfinite pixel map'); print(nf.sum(axis=0))
d=np.abs(np.where(np.isfinite(b),b,0)-a); print('max diff on finite', d.max()); bad=d[0]>2e-3; print('bad channels', np.nonzero(bad.any(axis=(1,2)))[0][:64]); print(bad.sum(axis=0))
"
