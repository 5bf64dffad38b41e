"""Dataset class-name tables (/root/reference/python/jdet/config/constant.py:167-223), DOTA family only."""
DOTA1_CLASSES = ['plane', 'baseball-diamond', 'bridge', 'ground-track-field', 'small-vehicle', 'large-vehicle', 'ship',
                 'tennis-court', 'basketball-court', 'storage-tank', 'soccer-ball-field', 'roundabout', 'harbor',
                 'swimming-pool', 'helicopter']
DOTA1_5_CLASSES = DOTA1_CLASSES + ['container-crane']
DOTA2_CLASSES = DOTA1_5_CLASSES + ['airport', 'helipad']


def get_classes_by_name(name):
    res = {'DOTA': DOTA1_CLASSES, 'DOTA1': DOTA1_CLASSES, 'DOTA1_5': DOTA1_5_CLASSES, 'DOTA2': DOTA2_CLASSES}
    assert name in res, name
    return res[name]
