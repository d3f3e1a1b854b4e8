set -x
python tools/ab_h3.py base spread4 spreadio spread4+spreadio --rounds=2 --iters=20 > gpurun_out/r02_ab3.txt 2>&1
cat gpurun_out/r02_ab3.txt
