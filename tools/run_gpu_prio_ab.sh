cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.Stream.priority_range())"
for rep in 1 2; do for pr in "" "-1"; do
SAT_MAIN_STREAM_PRIORITY=$pr timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mode > gpurun_out/prio_${pr}_$rep.json 2> gpurun_out/prio_${pr}_$rep.err || tail -3 gpurun_out/prio_${pr}_$rep.err
python -c "
import json
d=json.load(open('gpurun_out/prio_${pr}_$rep.json')); print('priority [$pr] rep $rep: %.0f img/s %.3f ms/step (min %.0f max %.0f)'%(d['value'],d['ms_per_step'],d['repeats']['images_per_sec_min'],d['repeats']['images_per_sec_max']))"
done; done
