for rep in 1 2; do
for set in "EVMI_WG_XCD=0" "EVMI_WG_XCD=1"; do
  a=$(env $set python3 bench.py --no-fs2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; p=json.loads(sys.stdin.read()); print(p['train']['ms_per_step'], p['train']['with_mrstft_loss']['ms_per_step'], p['ms_per_step'])")
  b=$(env $set python3 bench.py --no-fs2 --no-cpu-baseline --no-side-legs 2>/dev/null | python3 -c "import json,sys; p=json.loads(sys.stdin.read()); print(p['train']['ms_per_step'], p['train']['with_mrstft_loss']['ms_per_step'], p['ms_per_step'])")
  echo "$set | with side legs: GAN / +mrstft / infer ms = $a | without side legs: $b"
done
done
