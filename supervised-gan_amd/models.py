"""create_model (models/models.py:5-44): `--model` string -> trainer class."""


def create_model(opt):
    if opt.model == 'fcgan':
        from .fcgan_model import FCGANModel
        model = FCGANModel()
    elif opt.model == 'cgan':
        from .cgan_model import CGANModel
        model = CGANModel()
    elif opt.model == 'twostage_cycle':
        from .twostage_cycle_model import TwoStageCycleModel
        model = TwoStageCycleModel()
    elif opt.model == 'twostage':
        from .twostage_cycle_model import TwoStageModel
        model = TwoStageModel()
    elif opt.model == 'cgan2':
        from .cgan2_model import CGAN2Model
        model = CGAN2Model()
    elif opt.model == 'cgan_cycle':
        from .cgan_cycle_model import CGANCycleModel
        model = CGANCycleModel()
    elif opt.model == 'cgan2_cycle':
        from .cgan2_cycle_model import CGAN2CycleModel
        model = CGAN2CycleModel()
    elif opt.model in ('twostage_factd',
                       'test', 'segmentation', 'segmentation_cycle'):
        raise NotImplementedError("model [%s] is not on the MI355X path yet (fcgan, cgan, cgan2, cgan_cycle, cgan2_cycle, twostage and twostage_cycle are; see DESIGN.md scope)" % opt.model)
    else:
        raise ValueError("Model [%s] not recognized." % opt.model)
    model.initialize(opt)
    print("model [%s] was created" % (model.name()))
    return model
