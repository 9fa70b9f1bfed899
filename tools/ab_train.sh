q() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', d['value'], d['ms_per_step'], d['net_tflops'])"; }
for i in 1 2; do
python bench.py --mode train --steps 10 --warmup 2 --tune-cache gpurun_out/tc_train.json 2>/dev/null | q new
YOLO_LAB=1 YOLO_TRAIN_BN3=1 python bench.py --mode train --steps 10 --warmup 2 --tune-cache gpurun_out/tc_train.json 2>/dev/null | q bn3
done
