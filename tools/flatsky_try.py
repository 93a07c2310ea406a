import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cora_amd import _lib
from oracle import flatsky as ofs
ctx = _lib.get_context()
rng = np.random.default_rng(0)
worst = 0
for n in [1,2,3,4,5,7,8,12,16,31,32,64,100,128,257,1000,1024,2048,3000,4095,4096]:
    for shape, axis in [((3, n), 1), ((n, 5), 0), ((2, n, 37), 1)]:
        if np.prod(shape) > 4e6: continue
        x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        for inv in (False, True):
            d = torch.from_numpy(x.copy()).cuda()
            ctx.fft_c2c(d, axis, inverse=inv)
            ref = (np.fft.ifft if inv else np.fft.fft)(x, axis=axis)
            err = np.abs(d.cpu().numpy() - ref).max() / np.abs(ref).max()
            worst = max(worst, err)
            if err > 1e-12: print("c2c", n, shape, axis, inv, err)
print("c2c worst", worst)
worst = 0
for shape in [(8, 6, 10), (16, 16, 16), (30, 20, 14), (5, 7, 9), (128, 128), (1, 4), (100,), (4096,), (3, 250), (64, 64, 64), (17, 33, 50)]:
    x = rng.standard_normal(shape)
    s = np.fft.rfftn(x)
    d = ctx.rfftn(torch.from_numpy(x).cuda())
    err = np.abs(d.cpu().numpy() - s).max() / np.abs(s).max()
    sp = s + 0.3j * rng.standard_normal(s.shape)   # non-Hermitian DC/Nyquist imag parts are ignored like numpy
    ref = np.fft.irfftn(sp, s=shape)
    o = ctx.irfftn(torch.from_numpy(sp.copy()).cuda(), last=shape[-1])
    err2 = np.abs(o.cpu().numpy() - ref).max() / np.abs(ref).max()
    worst = max(worst, err, err2)
    print(shape, err, err2)
# partial axes
x = rng.standard_normal((5, 12, 9)) + 1j * rng.standard_normal((5, 12, 9))
ref = np.fft.irfft(np.fft.ifft(x, axis=1), axis=2)
o = ctx.irfftn(torch.from_numpy(x.copy()).cuda(), naxes=2)
print("partial", np.abs(o.cpu().numpy() - ref).max())
kw = rng.random((6, 8, 5))
sp = ctx.randomfield_draw(torch.from_numpy(kw).cuda(), 1234)
print("draw", np.abs(sp.cpu().numpy() - ofs.device_spec(kw, 1234)).max())
