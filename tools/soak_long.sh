#!/bin/bash
# long soak of the product loop (mimic_runner CLI, GHND Faster R-CNN b3ch, batch 16, 3x800x1333): N synthetic batches, loss and
# device time per iteration logged every 200; rc != 0 or a relay time-out anywhere fails.   usage (GPU box): bash tools/soak_long.sh 2000
N=${1:-2000}
mkdir -p gpurun_out/soak_long
python -m hnd_ghnd_object_detectors_amd.mimic_runner --config config/ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml \
  --json '{"teacher_model": {"backbone": {"params": {"pretrained": false}}, "params": {"pretrained": false}, "ckpt": "/tmp/none.pt"}, "student_model": {"backbone": {"params": {"pretrained": false}}, "params": {"pretrained": false}, "ckpt": "/tmp/soak_student.pt"}, "train": {"batch_size": 16, "log_freq": 200}}' \
  -distill --synthetic_batches $N --image_size 800x1333 --num_epochs 1 > gpurun_out/soak_long/soak.txt 2>&1
echo rc=$? >> gpurun_out/soak_long/soak.txt
grep -E "Epoch: \[0\]|device time|rc=|rror|time-out|timeout" gpurun_out/soak_long/soak.txt | tail -16
