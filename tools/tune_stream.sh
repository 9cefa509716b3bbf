cd $GRAFT_REPO_ROOT
for cfg in "12 3 8" "14 3 8" "12 4 8" "14 4 8" "12 2 8" "10 3 8" "12 3 6" "14 5 8"; do
  set -- $cfg
  v=$(timeout -k 10 120 python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline --acc-depth $1 --fold-threads $2 --products $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f %.4f %.4f' % (d['value'], d['ms_per_step'], d.get('steady_state_ms_per_step', 0)))")
  echo "lanes=$1 fold=$2 products=$3 -> $v"
done
