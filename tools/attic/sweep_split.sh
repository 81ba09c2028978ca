for sp in 1000,2000 40,80 30,60 20,40 14,28; do echo "split $sp"; U2MKD_TILE_SPLIT=$sp python tools/ab_tp.py 2>&1 | grep "64->64\|32->32" | cut -c1-150; done
