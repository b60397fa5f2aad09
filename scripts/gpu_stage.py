import os, sys, time
sys.path.insert(0, '.')
from cerebro_amd import capi
import numpy as np
D=4096
for rows in (4096,):
    for stop in (1,2,3,4,0):
        os.environ['CHIP_SCAN_DEBUG_STOP']=str(stop)
        chip=capi.Chip(D, capacity_hint=rows+200)
        chip.append_synthetic(rows+100, 1)
        chip.profile_enable(True)
        # use query_rows path repeatedly (sync) -- kernel time from events
        for i in range(5): chip.lib.chip_loop_reset(chip.h); 
        import ctypes as C
        p=capi.default_dot_params()
        chip.profile_reset()
        n=200
        for i in range(n):
            chip.loop_reset()
            try:
                chip.loop_tick_enqueue(rows+50, i%32, p)
            except Exception as e: print(e); break
            if i%32==31:
                chip.synchronize()
                for s in range(32):
                    try: chip.loop_tick_collect(s)
                    except Exception: pass
        chip.synchronize()
        ms,cnt,b,span=chip.profile_scan()
        print(f"rows={rows} stop={stop}: kernel avg {1e3*ms/max(cnt,1):.2f} us over {cnt}")
        chip.close()
