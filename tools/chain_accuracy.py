import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmm_amd import synth
from diffmm_amd.DataHandler import DiffusionData
from diffmm_amd.engine import DiffusionEngine
dev = torch.device("cuda:0")
torch.manual_seed(21)
g = synth.named("mid")
U, I, D, H, M, B = g.users, g.items, 64, 1024, 2, 300
for T_ in (40, 200):
    hp = dict(noise_scale=0.1, noise_min=0.0001, noise_max=0.02, sim_weight=0.1, reg=1e-5, lr=1e-3, seed=5)
    E = DiffusionEngine(I, D, H, 10, M, T_, hp, dev, B)
    for m in range(M):
        E.P[m].data.normal_(0, 0.04)
        E.P[m].view("in_layers.0.weight")[:, I + 10:].zero_()
    dd = DiffusionData(g.trn_rows, g.trn_cols, U, I, dev)
    users = torch.randperm(U, device=dev)[:B]
    x0 = dd.rows(users)
    lat = [v.clone() for v in E.generate_views_latent(xT=[x0] * M, B=B)]
    E.check_chain_error()
    E.hp["latent_fused"] = False; E._chain_plans.clear()
    per = [v.clone() for v in E.generate_views_latent(xT=[x0] * M, B=B)]
    E.hp["chain_state"] = "bf16"
    st = [v.clone() for v in E.generate_views_bf16(x0)]
    print("T", T_, "c1[:3]", E.c1[:3], "c1[-3:]", E.c1[-3:], "c2[-3:]", E.c2[-3:], "s_init", E._lat["s_init"], "coef max", float(E._lat["coef"].abs().max()))
    for m in range(M):
        ex = E.generate_view(m, x0, precision=3)
        sc = float(ex.abs().max())
        print(" m", m, "fused", float((lat[m]-ex).abs().max())/sc, "per-step", float((per[m]-ex).abs().max())/sc, "stepwise16", float((st[m]-ex).abs().max())/sc)
