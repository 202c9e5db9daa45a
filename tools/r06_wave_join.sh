#!/bin/bash
# A/B: how far into its stretch a lane looks for the join with its earlier pass (variants j256 .. j2048 with SVX_WPARSE_STATS)
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ds
for lv in "" "--level 6"; do for j in "$@"; do
  echo -n "join $j $lv: "; SVX_LIB=$PWD/build/libsvx_$j.so python3 tools/r06_wave_stats.py --dataset /tmp/ds $lv 2>&1 | tail -1 | python3 -c "
import sys,json; r=json.loads(sys.stdin.read()); p=r['per_member']; print('kernel %.2f ms  passes %.1f  windows %.1f  later_pass share %.3f  pass1 %.3f write %.3f  total_clk %.0f ok %s' % (r['kernel_ms'], p['passes'], p['windows'], r['share_of_total_clk']['later_pass_clk'], r['share_of_total_clk']['pass1_clk'], r['share_of_total_clk']['write_clk'], p['total_clk'], r['ok']))"
done; done
