"""Test-time output of the hot path: what the reference's testers hand to the
registration back end.

  * `test_record`   -- the dict IndoorTester.test dumps as `{snapshot_dir}/pth/{idx}.pth`
                       (ref:lib/tester.py:92-102): CPU tensors pcd / feats / overlaps / saliency, len_src, rot, trans;
  * `evaluate_pair` -- forward + MetricLoss feature-match recall of one pair (ref:lib/tester.py:48-83);
  * `probabilistic_sample` -- the overlap x saliency weighted sampling of interest points without
                       replacement that precedes RANSAC (ref:lib/tester.py:152-164).  It draws from the HOST numpy
                       generator exactly as the reference does (np.random.choice), so a seeded run picks the same
                       points; only the scores cross the bus (two [N] vectors).
RANSAC itself (open3d) is downstream of the path and out of scope."""
import numpy as np
import torch


def test_record(inputs, outputs):
    """ref:lib/tester.py:92-101."""
    return {
        "pcd": inputs["points"][0].detach().cpu(),
        "feats": outputs["feats_f"].detach().cpu(),
        "overlaps": outputs["scores_overlap"].detach().cpu(),
        "saliency": outputs["scores_saliency"].detach().cpu(),
        "len_src": int(inputs["stack_lengths"][0][0]),
        "rot": torch.as_tensor(inputs["rot"]).cpu(),
        "trans": torch.as_tensor(inputs["trans"]).cpu(),
    }


def evaluate_pair(model, desc_loss, inputs):
    """-> (record, stats): one iteration of IndoorTester.test's loop body without the file write."""
    with torch.no_grad():
        outputs = model(inputs)
        len_src = int(inputs["stack_lengths"][0][0])
        feats = outputs["feats_f"]
        loss_input = {
            "src_feats": feats[:len_src], "tgt_feats": feats[len_src:],
            "rot": inputs["rot"], "trans": inputs["trans"],
            "scores_overlap": outputs["scores_overlap"], "scores_saliency": outputs["scores_saliency"],
            "src_pcd_raw": inputs["src_pcd_raw"], "tgt_pcd_raw": inputs["tgt_pcd_raw"],
            "correspondences": inputs["correspondences"],
        }
        stats = desc_loss(loss_input)
    return test_record(inputs, outputs), stats


def probabilistic_sample(pcd, feats, scores, n_points):
    """Keep n_points rows drawn without replacement with probability proportional to `scores`
    (= overlap * saliency); clouds that are already small enough are returned unchanged
    (ref:lib/tester.py:152-164).  Returns (pcd, feats, idx) -- idx is None when nothing was dropped."""
    if pcd.shape[0] <= n_points:
        return pcd, feats, None
    s = scores.detach().cpu()
    probs = (s / s.sum()).numpy().flatten()
    idx = np.random.choice(np.arange(pcd.shape[0]), size=n_points, replace=False, p=probs)
    sel = torch.from_numpy(idx).to(pcd.device)
    return pcd[sel], feats[sel], idx
