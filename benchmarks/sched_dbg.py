import sys, numpy as np, torch
sys.path.insert(0, '.')
from dvd_amd import ops, schedule, synth
G = 32
x_t = synth.normalish("ss/x", (4, 2, G, G), 5); x0 = synth.uniform("ss/x0", (4, 2, G, G), -1, 1, 5)
tab = schedule.Tables(schedule.named_betas("cosine", 10))
f = np.float32
for i in (9, 5, 1):
    c = tab.ddim_coef(i)
    got = ops.sched_step(c, torch.from_numpy(x_t).cuda(), torch.from_numpy(x0).cuda()).cpu().numpy()
    eps = (f(c.c_recip) * x_t - x0) / f(c.c_recipm1)
    ref = x0 * f(c.sqrt_abar_prev) + f(c.dir_coef) * eps
    eps_r = (f(c.c_recip) * x_t - x0) * (f(1) / f(c.c_recipm1))
    ref_r = x0 * f(c.sqrt_abar_prev) + f(c.dir_coef) * eps_r
    d = got != ref
    print(i, 'mismatch', d.sum(), 'of', d.size, 'maxabs', np.abs(got - ref).max(), 'vs recip-variant mismatches', (got != ref_r).sum())
    idx = np.argwhere(d)[:3]
    for j in idx:
        j = tuple(j)
        print('   x_t', repr(x_t[j]), 'x0', repr(x0[j]), 'got', repr(got[j]), 'ref', repr(ref[j]), 'eps', repr(eps[j]))
