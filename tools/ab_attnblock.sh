for shape in "4096 1024 128" "4096 1024 64"; do
  echo "#### $shape"
  timeout -k 10 60 python tools/microbench.py attnblock $shape 2>&1 | grep -v amdgpu || exit 1
  for m in 16 32 64 128 160 240; do
    echo -n "ablate $m: "; DDPM_LIB=ablate_build/lib_attention_$m.so timeout -k 10 60 python tools/microbench.py attnblock $shape 2>&1 | grep -v amdgpu || exit 1
  done
  echo -n "projection alone: "; timeout -k 10 60 python tools/microbench.py conv1x1 $(echo $shape | awk '{print $1" "$2" 1 "$3" "$3+$3/4}') 2>&1 | grep -v amdgpu || exit 1
done
