for cfg in "8 8 0"; do
  set -- $cfg
  export TTI_GZ_MULT=$1 TTI_ST_MULT=$2 TTI_PAD=$3
  d=gpurun_out/tti3d_nt_$1_$2_$3
  timeout -k 10 100 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 tools/probe_tti3d.py 8 > $d.log 2>&1 || { echo failed $cfg; tail -5 $d.log; exit 1; }
  echo "== $cfg"; grep "TTI 3-D" $d.log; python3 tools/kstats.py $d tti3d
done
