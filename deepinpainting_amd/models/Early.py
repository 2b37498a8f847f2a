"""EarlyStopping — mirror of the reference's models/Early.py (train.ipynb cell 2: `early = EarlyStopping(20)`,
`early(valid_loss)`, `early.early_stop`).  Bookkeeping only."""


class EarlyStopping:
    """Raises `early_stop` once the validation loss has failed to improve `patience` times in a row (reference :4-21;
    an equal loss counts as an improvement there and here)."""

    def __init__(self, patience=8):
        self.patience = patience
        self.counter = 0
        self.best_score = None
        self.early_stop = False

    def __call__(self, val_loss):
        score = -val_loss
        if self.best_score is None or score >= self.best_score:
            self.best_score = score
            self.counter = 0
            return
        self.counter += 1
        if self.counter >= self.patience:
            self.early_stop = True
