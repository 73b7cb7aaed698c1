for d in 0 8 16 24 28; do GSA_DBG=$d python tools/dbg_time.py; done
