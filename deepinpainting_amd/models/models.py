"""create_model — the entry of the drop-in boundary.

Same contract as the reference's factory (models/models.py:2-12): `opt.model` selects the trainer class, the object comes
back initialised, an unknown name is a ValueError("Model [...] not recognized."), and the two progress lines the
notebooks show (`ipsr_net`, `model [IPSRModel] was created`) are printed.
"""
import importlib

# opt.model -> (module inside this package, class name); the reference knows exactly one trainer
_TRAINERS = {
    'ipsr_net': ('.IPSR', 'IPSR'),
}


def create_model(opt):
    from .. import use_shipped_miopen_db
    use_shipped_miopen_db()          # before the first convolution of this process
    wanted = opt.model
    print(wanted)
    entry = _TRAINERS.get(wanted)
    if entry is None:
        raise ValueError("Model [%s] not recognized." % wanted)
    trainer_cls = getattr(importlib.import_module(entry[0], __package__), entry[1])
    trainer = trainer_cls()
    trainer.initialize(opt)
    print("model [%s] was created" % trainer.name())
    return trainer
