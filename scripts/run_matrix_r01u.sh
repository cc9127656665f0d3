source scripts/bench_matrix.sh true
show() { python3 - <<PY
import json
ops=json.load(open("gpurun_out/ops_$1.json"))
print("   " + "  ".join("%s %.1f" % (o['name'].replace('backbone.','').replace('classifier.0','head'), o['ms']*1000) for o in ops if o['name'] in ('classifier.0','backbone.layer4.1.conv2','backbone.layer3.1.conv2','backbone.layer2.1.conv2','backbone.layer1.1.conv2')))
PY
}
run bf16_lat --steps 40 --warmup 5 --streams 1; show bf16_lat
NBC_CONV_KORDER=1 run bf16_lat_ko --steps 40 --warmup 5 --streams 1; show bf16_lat_ko
run bf16_s4 --steps 80 --warmup 8 --streams 4
NBC_CONV_KORDER=1 run bf16_s4_ko --steps 80 --warmup 8 --streams 4
run bf16_b8 --steps 10 --warmup 2 --streams 1 --batch 8; show bf16_b8
NBC_CONV_KORDER=1 run bf16_b8_ko --steps 10 --warmup 2 --streams 1 --batch 8; show bf16_b8_ko
