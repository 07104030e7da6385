"""`--model` string -> trainer (the reference's factory: models/models.py:5-44).  A table instead of an if-chain; names the
reference knows but this path does not carry raise NotImplementedError, unknown names ValueError (as the reference)."""
import importlib

_TRAINERS = {                       # --model         module                    class
    'fcgan':          ('fcgan_model',          'FCGANModel'),
    'cgan':           ('cgan_model',           'CGANModel'),
    'cgan2':          ('cgan2_model',          'CGAN2Model'),
    'cgan_cycle':     ('cgan_cycle_model',     'CGANCycleModel'),
    'cgan2_cycle':    ('cgan2_cycle_model',    'CGAN2CycleModel'),
    'twostage':       ('twostage_cycle_model', 'TwoStageModel'),
    'twostage_cycle': ('twostage_cycle_model', 'TwoStageCycleModel'),
    'twostage_factd': ('twostage_cycle_model', 'TwoStageFactDModel'),
    'segmentation':   ('segm_model',           'SegmentationModel'),
    'segmentation_cycle': ('segm_cycle_model', 'SegmentationCycleModel'),
    'test':           ('test_model',           'TestModel'),
}
_NOT_ON_THIS_PATH = ()


def create_model(opt):
    name = opt.model
    if name in _NOT_ON_THIS_PATH:
        raise NotImplementedError("model [%s] is not on the MI355X path yet (%s are; see DESIGN.md scope)"
                                  % (name, ', '.join(sorted(_TRAINERS))))
    if name not in _TRAINERS:
        raise ValueError("Model [%s] not recognized." % name)
    if name == 'test':
        assert opt.dataset_mode == 'single'      # models/models.py:31
    module, cls = _TRAINERS[name]
    model = getattr(importlib.import_module('.' + module, __package__), cls)()
    model.initialize(opt)
    print("model [%s] was created" % (model.name()))
    return model
