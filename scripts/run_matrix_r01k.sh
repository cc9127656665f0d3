source scripts/bench_matrix.sh true
show() { python3 - <<PY
import json
ops=json.load(open("gpurun_out/ops_$1.json"))
for o in ops:
    if o['name'] in ('classifier.0','backbone.layer4.1.conv2','backbone.layer4.1.conv1','backbone.layer4.1.conv3','backbone.layer3.1.conv2'):
        print("   %-28s %8.1f us" % (o['name'], o['ms']*1000))
PY
}
for t in 5 3; do
run bf16_t${t}_full --steps 20 --warmup 3 --streams 1 --conv-tile $t; show bf16_t${t}_full
NBC_CONV_ABLATE=1 run bf16_t${t}_nomfma --steps 20 --warmup 3 --streams 1 --conv-tile $t; show bf16_t${t}_nomfma
NBC_CONV_ABLATE=2 run bf16_t${t}_nodma --steps 20 --warmup 3 --streams 1 --conv-tile $t; show bf16_t${t}_nodma
done
run bf16_b8_t3_full --steps 6 --warmup 2 --streams 1 --conv-tile 3 --batch 8; show bf16_b8_t3_full
NBC_CONV_ABLATE=1 run bf16_b8_t3_nomfma --steps 6 --warmup 2 --streams 1 --conv-tile 3 --batch 8; show bf16_b8_t3_nomfma
NBC_CONV_ABLATE=2 run bf16_b8_t3_nodma --steps 6 --warmup 2 --streams 1 --conv-tile 3 --batch 8; show bf16_b8_t3_nodma
