import sys, os, ctypes, struct
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from dfq_amd import dfq, _ffi
from oracle import dfq_oracle as orc
g = np.load('/root/repo/tests/golden/kat_le_pairs.npz')
name='dw_pw'
arrs = {w: g['{}.in.{}'.format(name, w)].copy() for w in ('w1','w2','b1','bnw','bnb')}
t = {k: torch.from_numpy(v.copy()).cuda() for k,v in arrs.items()}
scum = torch.ones(24, device='cuda')
plan = dfq.LEPlan([(t['w1'], t['b1'], 1), (t['w2'], None, 1)], [(0, 1, t['bnw'], t['bnb'], scum)])
info = plan.level_info(0); print(info)
st = plan.trace(0, 0)
raw = struct.pack('16q', *st)
fl = struct.unpack('32f', raw)
print('thread0', fl[8:16]); print('thread1', fl[16:24])
a1 = arrs['w1'].reshape(24,-1); 
print('row0 min/max', a1[0].min(), a1[0].max(), 'col0 of w2', arrs['w2'].reshape(10,24)[:,0].min(), arrs['w2'].reshape(10,24)[:,0].max())
o = {k: v.copy() for k,v in arrs.items()}
S_o = orc.layer_equalization(o['w1'],o['w2'],o['b1'],o['bnw'],o['bnb'])
print('oracle S[:2]', S_o[:2])
t2 = {k: torch.from_numpy(v.copy()).cuda() for k,v in arrs.items()}
W1,W2,B1,S = dfq._layer_equalization(t2['w1'],t2['w2'],t2['b1'],t2['bnw'],t2['bnb'])
print('S gpu ', S.cpu().numpy()[:6]); print('S orc ', S_o[:6]); print('ratio', (S.cpu().numpy()/S_o)[:6])
print('w1 ratio', (t2['w1'].cpu().numpy().reshape(24,-1)/arrs['w1'].reshape(24,-1))[:3,:3])
print('b1 ratio', (t2['b1'].cpu().numpy()/arrs['b1'])[:6])
