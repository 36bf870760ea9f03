mkdir -p gpurun_out/r06l
python -m hnd_ghnd_object_detectors_amd.mimic_runner --config config/ghnd/faster_rcnn-backbone_resnet50-b3ch.yaml \
  --json '{"teacher_model": {"backbone": {"params": {"pretrained": false}}, "params": {"pretrained": false}, "ckpt": "/tmp/none.pt"}, "student_model": {"backbone": {"params": {"pretrained": false}}, "params": {"pretrained": false}, "ckpt": "/tmp/soak_student.pt"}, "train": {"batch_size": 16, "log_freq": 50}}' \
  -distill --synthetic_batches 400 --image_size 800x1333 --num_epochs 1 > gpurun_out/r06l/soak.txt 2>&1
echo rc=$? >> gpurun_out/r06l/soak.txt
grep -E "loss|device time|rc=" gpurun_out/r06l/soak.txt | tail -14
python tools/stress_gemm_variants.py --family native > gpurun_out/r06l/stress_native.txt 2>&1; tail -2 gpurun_out/r06l/stress_native.txt
python tools/stress_gemm_variants.py --family emulated > gpurun_out/r06l/stress_emulated.txt 2>&1; tail -2 gpurun_out/r06l/stress_emulated.txt
