# A/B harness for policy variants of the library on ONE GPU box (gpurun -- 'bash tools/variants_run.sh'): variants/lib_<name>.so are
# builds of csrc/ with different -D switches (variants/ is git-ignored but travels to the box); each is put in the library's place,
# runs tools/mcts_rounds.py (round trace) and the default bench line, and the original library is restored.  Edit the list below.
set -e
cp alphazeroforhnefatafl_amd/libtaflhip.so /tmp/lib_orig.so
for v in p0 p1 p2; do
  cp variants/lib_$v.so alphazeroforhnefatafl_amd/libtaflhip.so
  echo "== $v" >> gpurun_out/var.log
  python tools/mcts_rounds.py >> gpurun_out/var.log 2>&1
  python bench.py --no-cpu-baseline > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/var_$v.json')); print('$v', d['digest'][:420]); print('   executed', d['roofline']['playouts_executed'])" | tee -a gpurun_out/var.log
done
cp /tmp/lib_orig.so alphazeroforhnefatafl_amd/libtaflhip.so
