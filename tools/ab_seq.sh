for rep in 1 2; do
python tools/seq_sweep.py --seq 32,40 --out ab_old_$rep --env GGET_ATTN_BY_SAMPLE=0 2>&1 | tail -3
python tools/seq_sweep.py --seq 40 --out ab_noside_$rep --env GGET_ATTN_SIDE=0 2>&1 | tail -2
python tools/seq_sweep.py --seq 40,56 --out ab_side_$rep 2>&1 | tail -3
done
