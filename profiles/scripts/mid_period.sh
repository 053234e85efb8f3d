for p in 0 4 5; do echo "== NLH_QRX_PERIOD=$p"; NLH_QRX_PERIOD=$p python profiles/sweep_mid.py 4096x256:47,64,128,256 2048x128:128,256,512 2>&1 | grep batch; done
