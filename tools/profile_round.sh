#!/bin/bash
# Collect the round's evidence on the GPU box into gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/):
#   bench line + per-launch table, rocprofv3 kernel-trace summary of the same bench command, FETCH_SIZE / WRITE_SIZE of the
#   dominant kernel in two separate counter-only passes, isolated op timings.
# usage: tools/profile_round.sh r03
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf $out/kt
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o p -- python bench.py --no-cpu-baseline --no-inference --no-extras --steps 10 --warmup 3 > $out/kt.log 2>&1
db=$(ls $out/kt/*.db 2>/dev/null | head -1)
[ -n "$db" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-inference --no-extras --steps 10 --warmup 3   ($tag; 3 warm-up + 10 timed graph-replayed steps + 2 eagerly launched probe steps; durations include side-stream contention)"; python tools/rocpd_stats.py $db 70; } > $out/bench_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $out/pmc -o p -- python bench.py --no-cpu-baseline --no-probe --no-inference --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
  python - <<PY >> $out/pmc_traffic_raw.txt
import sqlite3, glob
c = sqlite3.connect(glob.glob('$out/pmc/*.db')[0])
for pat in ('conv3x3s1_kernel<unsigned short, 128, 64, 8, 1>', 'conv3x3_ws_kernel<64, false, 0, true, 0>', 'dcn_dom_bm_kernel<64, true>', 'dcn_dx_bm_kernel<2>', 'dcn_fwd_bm_kernel<2>', 'dcn_fwd_b2_kernel<false>', 'dcn_fwd_b2_kernel<true>', 'dcn_wgrad_bm_kernel', 'topk_map128_kernel', 'bn_bwd_apply_kernel<unsigned short', 'conv3x3_c16r_kernel<1, 1, 2>', 'conv3x3_c16r_kernel<1, 1, 0>', 'conv3x3_c16r_kernel<2, 2, 2>', 'stem7_rows_kernel', 'pack_weight_batch_kernel'):
    rows = c.execute("select dispatch_id, sum(value) from counters_collection where kernel_name like ? and counter_name='$c' group by dispatch_id", ('%' + pat + '%',)).fetchall()
    print('$c', pat, 'launches', len(rows), 'avg_kib', sum(r[1] for r in rows) / max(1, len(rows)))
PY
done
rm -rf $out/pmc
python tools/pmc_traffic_json.py $out/pmc_traffic_raw.txt > $out/pmc_traffic.json
# the bench line comes AFTER the counter passes: `roofline.traffic` is read from profiles/<tag>_pmc_traffic.json (stamped with the
# kernel source's blob hash), so the line of this run carries the traffic measured on this very tree
cp $out/pmc_traffic.json profiles/${tag}_pmc_traffic.json
[ -s $out/bench_kernel_stats.txt ] && cp $out/bench_kernel_stats.txt profiles/${tag}_bench_kernel_stats.txt    # roofline.rocprof of the line below reads it
[ -s profiles/${tag}_bench_kernel_stats.txt ] && python tools/trace_meta.py profiles/${tag}_bench_kernel_stats.txt && cp profiles/${tag}_bench_trace_meta.json $out/     # ... only with this sidecar (args + source hashes)
python bench.py --steps 20 --warmup 5 --probe-detail $out/ops_by_shape.txt > $out/bench.log 2>&1
tail -1 $out/bench.log > $out/bench_line.json
python tools/opbench.py decode bn conv > $out/opbench.txt 2>&1
DCN_SHAPES=3 python tools/opbench.py dcn >> $out/opbench.txt 2>&1
python tools/wgrad_bench.py 256 384 > $out/wgrad_bench.txt 2>&1
rm -rf $out/kt
# a REPLAYED step (--no-probe: every timed step of this trace is a graph replay): span, busy share, idle gaps, foreign kernels.  NOTE the
# profiler distorts the overlap of the two streams (profiles/<tag>_replay_timeline_unprofiled.txt is the unprofiled timeline)
timeout 600 rocprofv3 --kernel-trace -d $out/kt2 -o p -- python bench.py --no-cpu-baseline --no-inference --no-extras --no-probe --steps 10 --warmup 3 > $out/kt2.log 2>&1
db2=$(ls $out/kt2/*.db 2>/dev/null | head -1)
[ -n "$db2" ] && python tools/replay_trace.py $db2 > $out/replay_trace.txt 2>&1
rm -rf $out/kt2
python tools/zero_timeline.py > $out/replay_timeline_unprofiled.txt 2>&1
bash tools/pmc_sq.sh $tag > /dev/null 2>&1; cp gpurun_out/pmc_sq_$tag/pmc_sq.txt $out/pmc_sq.txt
python tools/bf16_error_growth.py train 16 256 > $out/bf16_error_growth_train.txt 2>&1
ls -la $out
