"""EarlyStopping — the patience counter train.ipynb drives (cell 2: `early = EarlyStopping(20)`, `early(valid_loss)`,
then `early.early_stop`); mirrors the behaviour of the reference's models/Early.py:4-21.  Bookkeeping only."""


class EarlyStopping:
    """Tracks the best (lowest) validation loss.  A loss that is not worse than the best resets the counter — equality
    counts as an improvement, as in the reference — and `patience` worse losses in a row set `early_stop`.
    `best_score` keeps the reference's sign convention (the negated loss)."""

    def __init__(self, patience=8):
        self.patience, self.counter = patience, 0
        self.best_score, self.early_stop = None, False

    def __call__(self, val_loss):
        candidate = -val_loss
        improved = self.best_score is None or candidate >= self.best_score
        if improved:
            self.best_score, self.counter = candidate, 0
        else:
            self.counter += 1
            self.early_stop = self.early_stop or self.counter >= self.patience
