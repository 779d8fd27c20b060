"""kNN-VC feature matching (reference module/tinyvc/feature_retrieval.py:15-33) on the streamed
top-k kernel of csrc/knn.hip."""
import torch

from ...engine import default_engine


def prepare_reference(reference):
    """Normalise + repack an index [1, 768, N] once.  The prepared blob rides on the tensor object
    itself (keyed by its in-place version counter), so it lives exactly as long as the index and can
    never be confused with another tensor that later reuses the same device address."""
    hit = getattr(reference, "_tvc_prepared", None)
    if hit is not None and hit[0] == reference._version and hit[1] == str(reference.device):
        return hit[2], hit[3]
    eng = default_engine(reference.device)
    blob, n = eng.knn_prepare(reference)
    try:
        reference._tvc_prepared = (reference._version, str(reference.device), blob, n)
    except Exception:
        pass
    return blob, n


@torch.no_grad()
def match_features(source, reference, k=4, alpha=0.0, metrics="cos", return_indices=False):
    """source [B, C, T], reference [B or 1, C, N] -> [B, C, T] (mean of the k nearest index vectors under `metrics` in
    {'cos', 'IP', 'L2'}, k = 1 ... 8, blended with the input by alpha): the reference's signature, feature_retrieval.py:15."""
    if reference.device != source.device:
        reference = reference.to(source.device)
    eng = default_engine(source.device)
    B = source.shape[0]
    if k != 4 or metrics != "cos":
        # every other argument of the reference's signature: plain fp32 on the raw index (csrc/knn_general.hip); the inference path's k = 4 /
        # 'cos' below runs the prepared-index search on the matrix pipe
        if reference.shape[0] not in (1, B):
            raise RuntimeError(f"batch of reference ({reference.shape[0]}) must be 1 or match source ({B})")
        if reference.shape[0] == 1:
            res = eng.knn_match_general(source, reference[0].float(), k, metrics, want_indices=True)
        else:
            parts = [eng.knn_match_general(source[b:b + 1], reference[b].float(), k, metrics, want_indices=True) for b in range(B)]
            res = (torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0))
        out, idx = res
        if alpha != 0.0:
            out = out * (1 - alpha) + source * alpha
        return (out, idx) if return_indices else out
    if reference.shape[0] == 1:
        blob, n = prepare_reference(reference)
        res = eng.knn_match(source, blob, n, want_indices=return_indices)
        out, idx = res if return_indices else (res, None)
    elif reference.shape[0] == B:
        outs, idxs = [], []
        for b in range(B):                       # one index per utterance
            blob, n = prepare_reference(reference[b:b + 1])
            res = eng.knn_match(source[b:b + 1], blob, n, want_indices=return_indices)
            o, i = res if return_indices else (res, None)
            outs.append(o)
            idxs.append(i)
        out = torch.cat(outs, 0)
        idx = torch.cat(idxs, 0) if return_indices else None
    else:
        raise RuntimeError(f"batch of reference ({reference.shape[0]}) must be 1 or match source ({B})")
    if alpha != 0.0:
        out = out * (1 - alpha) + source * alpha
    return (out, idx) if return_indices else out
