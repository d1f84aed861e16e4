# A/B of environment settings inside one box: r02_ab_env.sh "SIZES" "ENV1" "ENV2" ...   (ENV like A=1,B=2 or "-")
sizes=$1; shift
for e in "$@"; do
  echo "== $e"
  if [ "$e" = "-" ]; then python tools/enc_trunk_time.py $sizes --only-hip; else env $(echo $e | tr ',' ' ') python tools/enc_trunk_time.py $sizes --only-hip; fi
done
