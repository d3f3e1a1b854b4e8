"""Section timing of the split-fp16 net-block kernel from s_memtime stamps (tw_debug_set_flags bit 4 = 16):
wave 0 of workgroup 0 stamps the shader clock at the section boundaries of one coupling net."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow_oracle as fo
from tests import helpers as H
from timewarp_amd import _lib

args = [a for a in sys.argv[1:] if not a.startswith("--")]
extra = int(args[0]) if args else 0
H1 = "--h1" in sys.argv         # the single-MFMA build (TW_PATH_FUSED_H1): encoder-stack statement with its stamps always compiled in
DENSE = "--dense" in sys.argv   # the split-fp16 dense-softmax kernel (transformer_nvp) instead of the kernel-attention one
sd = H.full_dense_sd() if DENSE else H.full_kernel_sd()
N, V = 1000, 22
for a in sys.argv[1:]:          # --atoms=60 --rows=512: e.g. the 64-token build (flag 65536) at BASELINE config 3's size
    if a.startswith("--atoms="):
        V = int(a.split("=")[1])
    if a.startswith("--rows="):
        N = int(a.split("=")[1])
g = torch.Generator().manual_seed(2)
at = torch.randint(0, 5, (1, V), generator=g)
x_c = torch.randn(1, V, 3, generator=g) * 0.3
x_v = torch.randn(1, V, 3, generator=g) * 0.5
zo = torch.randn(N, V, 3, generator=g) * 0.5
mask = torch.zeros(1, V, dtype=torch.bool)
xc = x_c - fo.centre_of_mass(x_c, mask)
PATH = 4 if H1 else 3
if H1:
    extra |= 8192
m = H.tw_dense_model(sd, path=3) if DENSE else H.tw_kernel_model(sd, path=PATH)
_lib.load().tw_debug_set_flags(16 | extra)
for rep in range(3):
    acts, out = m.debug_netblock(0, 0, at.cuda(), xc.cuda(), x_v.cuda(), mask.cuda(), zo.cuda(), PATH)
torch.cuda.synchronize()
ts = acts.reshape(-1)[:160].contiguous().view(torch.int64).cpu().tolist()
L = 3
names = ["start", "in_mlp"]
for l in range(L):
    names += [f"L{l} attention", f"L{l} add+LN1", f"L{l} FFN", f"L{l} add+LN2"]
names += ["out_mlp"]
t = ts[: len(names)]
total = t[-1] - t[0]
print(f"total {total} cycles")
if ts[60]:
    print(f"  prologue in front of the first stamp (token bookkeeping, embedding gather, previous coupling update, first weight "
          f"stages): {t[0] - ts[60]} cycles")
    if ts[64]:
        print(f"    (fine: weight-stage requests {ts[64] - ts[60]}, token loop + load issue {ts[65] - ts[64]}, coupling update {ts[66] - ts[65]}, log-det {ts[61] - ts[66]})")
    if ts[61]:
        print(f"    token bookkeeping + z loads {ts[61] - ts[60]}, input features {ts[62] - ts[61]}, LDS zeroing + bias loads {ts[63] - ts[62]}, "
              f"wait for the first stages {t[0] - ts[63]}")
if extra & 8192:   # encoder-stack build (tools/gen_h3_enc_asm.py); inner stamps only from its H3_ENC_EXPERIMENT=stamps build
    print(f"  in_mlp {t[1] - t[0]}, encoder stack {t[-2] - t[1]} ({(t[-2] - t[1]) // L} per layer), out_mlp {t[-1] - t[-2]}")
    prev = t[1]
    for l in range(L):
        a0, a1, f0, f1 = ts[40 + 4 * l: 44 + 4 * l]
        end = ts[2 + 4 * l + 3]
        if a0 and a1 > a0:
            print(f"  layer {l}: before the layer {a0 - prev}, barrier + side DMA + attention {a1 - a0}, LN1 glue {f0 - a1}, FFN {f1 - f0}, "
                  f"LN2 glue (+ transposer) {end - f1}")
        prev = end
    sys.exit(0)
agg = {}
for i in range(1, len(names)):
    d = t[i] - t[i - 1]
    key = names[i].split(" ", 1)[-1] if names[i].startswith("L") else names[i]
    agg[key] = agg.get(key, 0) + d
    print(f"  {names[i]:16s} {d:9d}  {100.0 * d / total:5.1f} %")
print({k: f"{100.0 * v / total:.1f}%" for k, v in agg.items()})

# finer split of one layer (asm build only): glue code around the two asm blocks
l = 1
a0, a1, f0, f1 = ts[40 + 4 * l: 44 + 4 * l]
if a0 and a1 > a0:
    prev = t[1 + 4 * l]            # end of layer l-1 (or in_mlp)
    print(f"layer {l}: x^T write {a0 - prev}, attention asm {a1 - a0}, y readback+scale {t[2 + 4 * l] - a1}, "
          f"LN1 {t[3 + 4 * l] - t[2 + 4 * l]}, split+xb write {f0 - t[3 + 4 * l]}, FFN asm {f1 - f0}, "
          f"y readback+bias {t[4 + 4 * l] - f1}, LN2 {t[5 + 4 * l] - t[4 + 4 * l]}")
