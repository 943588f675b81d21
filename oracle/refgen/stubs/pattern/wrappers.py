class VisPattern:
    pass
