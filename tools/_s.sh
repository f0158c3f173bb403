for aff in 0 1; do for st in 0 1; do for res in 0 2; do
CB_AFF=$aff CB_STATS=$st CB_RES=$res python tools/conv_bench.py 40 256 768 16 32 1 $aff 0 2>&1 | tail -1
done; done; done
