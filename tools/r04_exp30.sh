for lib in - tools/variants/libgpar_ratio4.0.so tools/variants/libgpar_ratio0.0.so; do echo "== $lib"; python tools/r04_marginal_potrf.py $lib 2>&1 | grep -v amdgpu; done
