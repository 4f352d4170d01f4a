for g in graph eager; do for m in overlap single; do
  if [ $g = eager ]; then extra="--no-graph"; else extra=""; fi
  python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --profile-steps 0 --dp-dry-run $m $extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('mode=$m', '$g', d['ms_per_step'])"
done; done
