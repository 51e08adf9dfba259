# flow3d/train.py at the bench's size on its three data paths: device-generated triplets, the training set
# resident in pinned host memory behind the side-stream prefetcher (--host_data --host_cache), each under the default step
# driver (HIP-graph replay) and with --eager
mkdir -p gpurun_out
O=gpurun_out/train256.txt
: > $O
for extra in "" "--eager" "--host_data --host_cache" "--host_data --host_cache --eager"; do
  echo "## python -m opticalflowscivis_amd.flow3d.train --size 256 --batch_size 2 --samples 24 --epoch 2 $extra" >> $O
  timeout -k 10 500 python -m opticalflowscivis_amd.flow3d.train --dataset droplet3d --size 256 --samples 24 --batch_size 2 --epoch 2 --mode train \
      --log_every 6 --log_path /tmp/tl256 $extra 2>&1 | grep "train loop\|eval epoch" >> $O
  rm -rf /tmp/tl256
done
cat $O
