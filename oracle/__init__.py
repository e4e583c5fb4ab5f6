"""CPU oracle for the masked-diffusion hot path.  TEST INFRASTRUCTURE ONLY.

This package is an own-words CPU (torch fp32 / float64) restatement of the
reference's arithmetic for the path named in BASELINE.json `north_star`:

  scheduler_ref.py  <- code/scheduler.py   (schedules, degrade, shift, weights)
  unet_ref.py       <- code/models/unet/unet6.py + models_Unet.py:132-171
  sampler_ref.py    <- code/sampler.py:46-83, 109-261
  trainer_ref.py    <- code/trainer_masked_mean_shift.py:82-193,
                       code/trainer_masked.py:95-183

Every function cites the reference file:line it follows.  The oracle is pinned
against golden vectors produced by importing the reference itself in the build
container (tests/golden/make_golden.py; fixtures in tests/golden/*.npz).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (masked-diffusion-model_amd/mdm) never does: it fails
loudly when the HIP extension is missing.

There is no C restatement here (the reference is pure Python/torch floating
point), so __graft_entry__.build() has nothing to compile under oracle/.
"""
