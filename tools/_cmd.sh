for p in -1 1 0; do echo FB_PAIR=$p; MIFLOW_FB_PAIR=$p timeout 100 python tools/fb_single.py 640 480 400; done
