"""Data-source helpers -- counterpart of gpmp/kernel/utils.py."""
from .. import num as gnp


def check_xi_zi_or_loader(xi, zi, dataloader):
    """utils.py:12-20: exactly one of (xi, zi) / dataloader -> "arrays" | "dataloader"."""
    arrays = xi is not None and zi is not None
    if arrays and dataloader is not None:
        raise ValueError("Provide either (xi, zi) or dataloader, not both.")
    if not arrays and dataloader is None:
        raise ValueError("Provide either (xi, zi) or dataloader.")
    return "arrays" if arrays else "dataloader"


def prepare_data(xi=None, zi=None, loader=None):
    """utils.py:22-38: (xi, zi column, n, d, source) from arrays, or (None, None, n, d, "loader") from a loader whose
    dataset exposes ``x_list`` as the reference's DataLoader does."""
    arrays = xi is not None and zi is not None
    if arrays and loader is not None:
        raise ValueError("Provide either (xi, zi) or loader, not both.")
    if not arrays and loader is None:
        raise ValueError("Provide either (xi, zi) or loader.")
    if arrays:
        xi_, zi_ = gnp.asarray(xi), gnp.asarray(zi).reshape(-1, 1)
        n, d = xi_.shape
        return xi_, zi_, n, d, "arrays"
    return None, None, len(loader.dataset), loader.dataset.x_list[0].shape[1], "loader"
