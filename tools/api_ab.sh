for v in off auto off auto; do
  YOLO_CU_PARTITION=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['config']['detect_api_images_per_s'])"
done
