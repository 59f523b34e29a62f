"""CPU oracle for the BitDance next-patch-diffusion generation path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``bitdance_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / reported baseline.

What it is: a CPU (torch-CPU / numpy) restatement of the arithmetic the
reference executes on the hot path named in SURVEY.md section 8:

  * ``sampler``    -- Euler-Maruyama SDE sampler (sampling_x.py)
  * ``diff_head``  -- binary diffusion head / adaLN DiT (flow_head_parallel_x.py)
  * ``qwen3``      -- decoder-only LLM forward as the reference drives it
                     (HF transformers Qwen3Model, third party, pinned 4.57.0 by
                     requirements.txt:1; restated from the published algorithm)
  * ``pipeline``   -- AR orchestration, pos-embed, projector, sign binarise
                     (t2i_pipeline.py)
  * ``gfq``        -- bit <-> index math of the group-wise LFQ (imagenet_gen/src/gfq.py)
  * ``autoencoder`` -- the binary tokenizer's conv encoder / decoder
                     (vision_encoder/autoencoder.py), functional from the state dict,
                     fp32 and autocast policies; pinned against ae_roundtrip.npz.  The
                     native conv kernels (csrc/bd_conv.hip) are checked against it.

Parity pinning: the reference repository ships no tests or golden vectors
(SURVEY.md section 4), so the oracle is pinned against the reference *itself*:
``oracle/gen_golden.py`` imports the unmodified reference from /root/reference
(with the three version shims of SURVEY.md section 8c), runs it on seeded
random weights with injected noise and stores inputs/outputs under
``tests/golden/``.  ``tests/test_oracle_golden.py`` replays those vectors
through this restatement (no /root/reference needed at test time).

Every function cites the reference file:line it follows.
"""
