import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpe_amd
from gpe_amd import ops
from oracle import ref_path as O
B, N, C, ld, k = 2, 300, 150, 152, 16
g = torch.Generator().manual_seed(B * 7 + N + C + k)
torch.randn(B * N, C, generator=g) if False else None
cen = torch.randn(8, C, generator=g) * 20
lab = torch.randint(0, 8, (B * N,), generator=g)
x = cen[lab] + 1e-2 * torch.randn(B * N, C, generator=g)
buf = torch.zeros(B * N, ld); buf[:, :C] = x
ref = O.knn_local(x.contiguous(), B, k).to(torch.int32).view(B, N, k)
got = ops.knn(buf.cuda()[:, :C], B, N, k).cpu()
bad = (got != ref).any(-1)
print('bad', bad.sum().item())
cnt = torch.bincount(lab[:N], minlength=8), torch.bincount(lab[N:], minlength=8)
print('cluster sizes', cnt)
nb = 0
for b in range(B):
    for i in range(N):
        if bad[b, i] and nb < 6:
            nb += 1
            print('q', b, i, 'cluster size', cnt[b][lab[b * N + i]].item())
            print('  got', got[b, i].tolist())
            print('  ref', ref[b, i].tolist())
            d = O.sqdist_one_cloud(x[b * N:(b + 1) * N])[i]
            print('  d got', [round(d[j].item(), 5) for j in got[b, i].tolist()])
            print('  d ref', [round(d[j].item(), 5) for j in ref[b, i].tolist()])
sizes = torch.stack([cnt[b][lab[b * N:(b + 1) * N]] for b in range(B)])
print('bad by cluster size:', sorted(set(sizes[bad].tolist())), ' ok sizes:', sorted(set(sizes[~bad].tolist())))
import struct
idx, jg = ops.knn(buf.cuda()[:, :C], B, N, k, want_global=True)
jg = jg.cpu()
def f(v): return struct.unpack('f', struct.pack('i', int(v)))[0]
for (b, i) in [(0, 21), (0, 29), (0, 22)]:
    r = jg[b, i].tolist()
    print('q', b, i, 'm', r[0], 'fl %08x' % (r[1] & 0xffffffff), 'm2e', f(r[2]), 'da0', f(r[3]), 'da1', f(r[4]), 'T', f(r[5]), 'dex0', f(r[6]), 'dex1', f(r[7]), 'r0', r[8], 'r1', r[9], 's0', r[10], 'id0', r[11], 'newpos0', r[12])
