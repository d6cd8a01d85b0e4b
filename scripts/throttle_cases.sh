# cases for scripts/throttle_matrix.sh <outdir> scripts/throttle_cases.sh: driver wait mode x worker count x slots
run poll_default MCORB_HOST_PROF=1
run spin MCORB_SYNC=spin
run poll_ht10 MCORB_HOST_THREADS=10
run poll_ht14 MCORB_HOST_THREADS=14
run poll_default2 MCORB_X=1
run spin2 MCORB_SYNC=spin
BARGS="--slots 8"
run s8_poll MCORB_X=1
BARGS="--slots 10"
run s10_poll MCORB_X=1
