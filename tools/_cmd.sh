for v in 4 7 10 14; do echo FB_MODEL=$v; for sz in "1920 1080" "1280 720" "640 480"; do MIFLOW_TILE_FB_MODEL=$v timeout 200 python tools/tvl1_single.py $sz 30 | cut -c1-80; done; done
