for c in 0 16 32 64 96 128 256; do for nt in 1 0; do NLH_FD_CHUNK_MB=$c NLH_FDQ_NT=$nt python profiles/scripts/devfcn_fd.py 512 4096 256 2>&1 | grep chunk; done; done
NLH_FD_CHUNK_MB=64 python profiles/scripts/devfcn_fd.py 2048 2048 128 2>&1 | grep chunk
NLH_FD_CHUNK_MB=0 python profiles/scripts/devfcn_fd.py 2048 2048 128 2>&1 | grep chunk
