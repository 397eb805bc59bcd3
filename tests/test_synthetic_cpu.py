"""Host-side synthetic inputs (vila_amd/synthetic.py)."""
import torch

from vila_amd import configs, synthetic


def test_make_prompt_never_draws_a_media_or_eos_id_whatever_the_id_order():
    """`_media_plan` registers the video token as well as the image token, so a random text id equal to `video_token_id` would
    raise without a video.  The committed fixtures' configs keep the video id outside the drawn range (their draws must not move); a config
    whose video id lies below its image id gets those draws moved off it."""
    cfg = configs.tiny("mlp_downsample")
    base = synthetic.make_prompt(cfg, 4000, 1, 3)
    assert int(base[0]) == cfg.image_token_id
    for bad in (cfg.image_token_id, cfg.video_token_id, cfg.llm.eos_token_id):
        assert not bool((base[1:] == bad).any())
    cfg2 = configs.tiny("mlp_downsample")
    cfg2.video_token_id = 500                                  # inside the drawn range
    moved = synthetic.make_prompt(cfg2, 4000, 1, 3)
    assert bool((base[1:] == 500).any()) and not bool((moved[1:] == 500).any())
    assert torch.equal(moved[base != 500], base[base != 500])  # every other draw is untouched
