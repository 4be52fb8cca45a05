import torch, sys
sys.path.insert(0, '/root/repo')
from desco_amd import ops
DEV='cuda'
def run(S, sm, num_rows, deg, wmode):
    g = torch.Generator().manual_seed(1)
    n_all = num_rows
    x = torch.randn(n_all, 64, generator=g)
    cnt = torch.full((n_all*S,), deg, dtype=torch.long)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(cnt, 0)]).to(torch.int32)
    col = torch.randint(0, n_all, (int(ptr[-1]),), generator=g).to(torch.int32)
    wt = torch.zeros((sm+1)*64, 64)
    # weight block b = identity * (b+1) if wmode == 'id' else random
    if wmode == 'id':
        for b in range(sm+1): wt[b*64:(b+1)*64] = torch.eye(64)*(b+1)
    else:
        wt = torch.randn((sm+1)*64, 64, generator=g)/8
    bias = torch.zeros(64)
    agg = torch.zeros(n_all*S, 64, dtype=torch.double)
    agg.index_add_(0, torch.repeat_interleave(torch.arange(n_all*S), cnt), x.double()[col.long()])
    A = torch.cat([agg.view(n_all, S*64)[:, :sm*64], x.double()], 1)
    ref = torch.relu(A @ wt.double())
    out = torch.full((n_all, 64), -7.0, device=DEV)
    ops.shmp_layer(x.to(DEV), ptr.to(DEV), col.to(DEV), 0, num_rows, S, sm, wt.to(DEV), bias.to(DEV), out)
    d = (out.cpu().double()-ref).abs()
    print(f"S={S} sm={sm} rows={num_rows} deg={deg} {wmode}: maxdiff {d.max().item():.3e} bad rows {(d.max(1).values>1e-3).sum().item()} badcols {(d.max(0).values>1e-3).nonzero().flatten().tolist()[:12]}")
for args in [(4,0,1,0,'id'),(4,0,1,0,'rnd'),(4,0,40,0,'rnd'),(4,1,1,1,'id'),(4,1,1,1,'rnd'),(4,2,40,2,'rnd'),(4,3,1,1,'id'),(4,3,1,1,'rnd'),(4,3,400,2,'rnd'),(2,2,400,2,'rnd'),(4,2,5000,2,'rnd')]:
    run(*args)
