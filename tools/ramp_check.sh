for args in "--steps 20 --warmup 5" "--steps 20 --warmup 100" "--steps 200 --warmup 20" "--steps 20 --warmup 5 --static-camera" "--steps 200 --warmup 20 --static-camera"; do
  timeout -k 10 200 python3 bench.py --gpus 1 --no-cpu-baseline $args | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args:', round(d['value'], 1), 'frames/s')"
done
