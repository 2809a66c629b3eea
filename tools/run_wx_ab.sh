R=$PWD; cd /tmp && export TMPDIR=/tmp
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
for X in 1 0; do
  export GLAM_WGRAD_X3=$X
  rm -rf /tmp/f$X /tmp/w$X
  rocprofv3 --pmc FETCH_SIZE -d /tmp/f$X -o f -- python3 $R/tools/wx_time.py 326400 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/w$X -o w -- python3 $R/tools/wx_time.py 326400 > /dev/null 2>&1
  echo "== GLAM_WGRAD_X3=$X"; python3 $R/tools/rocpd_traffic.py $(db /tmp/f$X) $(db /tmp/w$X) | head -4
done
