class DreamTrainer:
    pass
