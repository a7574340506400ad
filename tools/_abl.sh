for m in h0 h1 h0rb2 h0l0; do echo "VAR=$m"; DKT_LIB_PATH=$PWD/dkt_stereo_amd/lib/variants/lib_$m.so timeout 100 python tools/conv_ws_check.py --time 2>&1 | grep "us per launch\|worst"; done
