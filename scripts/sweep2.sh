for pad in 0 16384 24576 36864 65536 131072; do
echo -n "pad=$pad "; ANDI_LDS_PAD=$pad bash scripts/sweep.sh "8" "${1:-4096}" "12"
done
