python tools/ab_h3.py base attn:nobarrier attn:novalu attn:noxt attn:nosf attn:nomix --rounds=2 --iters=20 > gpurun_out/r02_ab_attn.txt 2>&1
cat gpurun_out/r02_ab_attn.txt
