import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer, Criterion
from musicgeneration_amd.optim import FusedAdamW
torch.manual_seed(0)
m = AlibiTransformer(vocab=618, embed_dim=64, n_layers=2, n_heads=8, max_sequence=64, proj_forward=128, dropout=0.0, dtype="bf16").cuda()
opt = FusedAdamW(m.parameters(), lr=1e-2)
x = torch.randint(0, 618, (2, 48)).cuda()
w = m.decoder_layers[0].attn.c_attn.w
for it in range(3):
    loss, _ = Criterion(1)(m(x), torch.roll(x, -1, 1)); opt.zero_grad(); loss.backward()
    v0 = w._version; opt.step(); print(it, float(loss), "version", v0, "->", w._version)
    W = m.decoder_layers[0].weights(torch.bfloat16)
    print("   shadow max diff vs master:", float((W.wqkv.float() - w.detach().view(64, 192)).abs().max()))
